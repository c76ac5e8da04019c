// Micro-benchmark (development aid): do f32 MFMAs of one wave overlap with f32/int VALU work of
// another wave on the same SIMD?  Blocks of 512 threads = 8 waves = 2 per SIMD.  mode 0: all
// waves MFMA; 1: all waves VALU; 2: first 4 waves MFMA, last 4 VALU (one of each per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VOP>
__global__ __launch_bounds__(512) void k(float *out, int n, int mode)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = lane + 4, x5 = lane + 5, x6 = lane + 6, x7 = lane + 7;
    float b = 1.0001f;
    if (do_mfma) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, b, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, b, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x3, b, a3, 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {     // 128 VALU per iteration (vs 16 MFMA = 512 cycles)
                if (VOP == 0)
                    asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                                 "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
                else
                    asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                                 "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int VOP>
float run(float *d, int blocks, int n, int mode)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VOP>, dim3(blocks), dim3(512), 0, 0, d, 4, mode);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<VOP>, dim3(blocks), dim3(512), 0, 0, d, n, mode);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float *d; (void)hipMalloc(&d, sizeof(float) * 512 * ncu);
    const int n = 2000;
    for (int vop = 0; vop < 2; ++vop) {
        float t0 = vop ? run<1>(d, ncu, n, 0) : run<0>(d, ncu, n, 0);
        float t1 = vop ? run<1>(d, ncu, n, 1) : run<0>(d, ncu, n, 1);
        float t2 = vop ? run<1>(d, ncu, n, 2) : run<0>(d, ncu, n, 2);
        printf("%s: all-MFMA %.3f ms (2 waves/SIMD x %d MFMA)   all-VALU %.3f ms (2 waves/SIMD x %d VALU)   1 MFMA wave + 1 VALU wave per SIMD %.3f ms\n",
               vop ? "v_max_f32" : "v_add_f32", t0, 16 * n, t1, 128 * n, t2);
        printf("   expected if they overlap fully: %.3f ms; if they serialize: %.3f ms\n", (t0 > t1 ? t0 : t1) / 2, (t0 + t1) / 2);
    }
    return 0;
}
