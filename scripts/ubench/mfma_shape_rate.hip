// Micro-benchmark (development aid): what does the chip sustain, under its power budget, on the two shapes of the 16-bit MFMAs --
// v_mfma_f32_16x16x32_{bf16,f16} (what the EarlyFusion GEMMs issue) against v_mfma_f32_32x32x16_{bf16,f16} -- with a wave's 64 x 64
// cells in 64 accumulator registers, random operands (two operand sets alternating, as a k loop's chunks do), two waves per SIMD,
// one workgroup per CU?  Prints ms, TFLOP/s and the effective clock (s_memtime ticks of a workgroup / wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, bool F16>
__global__ __launch_bounds__(512) void k(const uint4 *__restrict__ src, float *out, unsigned long long *ticks, int n)
{
    const int lane = threadIdx.x & 63;
    uint4 A[2][4], B[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            A[s][q] = src[(threadIdx.x * 16 + s * 8 + q) % 8192];
            B[s][q] = src[(threadIdx.x * 16 + s * 8 + 4 + q + blockIdx.x) % 8192];
        }
    const unsigned long long t0 = __builtin_readcyclecounter();
    float sum = 0.0f;
    if (SHAPE == 16) {
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (F16) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, B[s][b]), __builtin_bit_cast(f16x8, A[s][a]), acc[a][b], 0, 0, 0);
                        else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, B[s][b]), __builtin_bit_cast(bf16x8, A[s][a]), acc[a][b], 0, 0, 0);
                    }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            if (F16) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, B[s][2 * b + kh]), __builtin_bit_cast(f16x8, A[s][2 * a + kh]), acc[a][b], 0, 0, 0);
                            else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, B[s][2 * b + kh]), __builtin_bit_cast(bf16x8, A[s][2 * a + kh]), acc[a][b], 0, 0, 0);
                        }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) sum += acc[a][b][i];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = sum;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    (void)lane;
}

template <int SHAPE, bool F16>
void run(const uint4 *src, float *d, unsigned long long *dt, int blocks, int n, const char *name)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(512), 0, 0, src, d, dt, n / 8);     // warm the clocks
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(512), 0, 0, src, d, dt, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> t((size_t)blocks);
        (void)hipMemcpy(t.data(), dt, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : t) mean += (double)v; mean /= blocks;
        const double flops = 2.0 * 64 * 64 * 64 * (double)n * 8 * blocks;       // per iteration and wave: two chunks of 64 x 64 x 32
        printf("%-28s %8.3f ms  %7.1f TFLOP/s  ticks per workgroup %.4g (%.3f GHz if a tick is a shader cycle; 100 MHz if it is the constant clock)  cycles per 8192-MAC unit and SIMD %.2f\n",
               name, ms, flops / ms * 1e-9, mean, mean / (ms * 1e6), mean / ((double)n * 2 * 16 * 2));
    }
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    const int zero = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<unsigned short> hb(8192 * 8), hf(8192 * 8);
    srand(7);
    for (size_t i = 0; i < hb.size(); ++i) {
        const float v = zero ? 0.0f : (float)rand() / RAND_MAX * 2.0f - 1.0f;
        unsigned u; memcpy(&u, &v, 4);
        hb[i] = (unsigned short)(u >> 16);
        const _Float16 h = (_Float16)v; memcpy(&hf[i], &h, 2);
    }
    uint4 *sb, *sf; float *d; unsigned long long *dt;
    (void)hipMalloc(&sb, hb.size() * 2); (void)hipMalloc(&sf, hf.size() * 2);
    (void)hipMemcpy(sb, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(sf, hf.data(), hf.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&d, sizeof(float) * 512 * ncu); (void)hipMalloc(&dt, 8 * ncu);
    const int n = 40000;
    printf("%d CUs, %s operands\n", ncu, zero ? "ZERO" : "random");
    run<16, false>(sb, d, dt, ncu, n, "16x16x32 bf16");
    run<32, false>(sb, d, dt, ncu, n, "32x32x16 bf16");
    run<16, true>(sf, d, dt, ncu, n, "16x16x32 f16");
    run<32, true>(sf, d, dt, ncu, n, "32x32x16 f16");
    run<16, false>(sb, d, dt, ncu, n, "16x16x32 bf16 (again)");
    run<32, false>(sb, d, dt, ncu, n, "32x32x16 bf16 (again)");
    return 0;
}
