// Micro-benchmark (development aid): does LDS traffic of one wave overlap with VALU / MFMA work of
// another wave on the same SIMD (and CU)?  Blocks of 512 threads = 8 waves = 2 per SIMD.
// kind A (waves 0-3) and kind B (waves 4-7): 0 = ds_read_b32 stream, 1 = v_add_f32 stream,
// 2 = f32 MFMA stream, 3 = ds_write_b128 stream, 4 = ds_add_u32 random.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void work(int kind, int n, float &acc, float *lds, int lane)
{
    float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
    const float b = 1.0001f;
    const unsigned la = (unsigned)(size_t)lds + 4u * lane;
    if (kind == 0) {
        unsigned t0, t1, t2, t3;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                             : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(la) : "memory");
                x0 += __uint_as_float((t0 ^ t1 ^ t2 ^ t3) & 1u);
            }
        }
    } else if (kind == 1) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
                             "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));
        }
    } else if (kind == 2) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, b, a1, 0, 0, 0);
            }
        }
        x0 += a0[0] + a1[1];
    } else if (kind == 5) {      // bf16 MFMA 16x16x32 (matrix pipe)
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 av, bw;
        for (int q = 0; q < 8; ++q) { av[q] = (__bf16)(x0 + q); bw[q] = (__bf16)(x1 - q); }
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bw, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw, av, a1, 0, 0, 0);
            }
        }
        x0 += a0[0] + a1[1];
    } else if (kind == 3) {
        const unsigned lw = (unsigned)(size_t)lds + 16u * lane;
        f32x4 v = {x0, x1, x2, x3};
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n" :: "v"(lw), "v"(v) : "memory");
        }
    } else {
        unsigned addr = (unsigned)(size_t)lds + (((unsigned)lane * 2654435761u) >> 23) * 4u;
        unsigned one = 1u;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:36\n ds_add_u32 %0, %1 offset:72\n ds_add_u32 %0, %1 offset:108\n" :: "v"(addr), "v"(one) : "memory");
                addr = (unsigned)(size_t)lds + ((addr * 5u + 4u * 77u) & 2044u);
            }
        }
    }
    acc += x0 + x1 + x2 + x3;
}

__global__ __launch_bounds__(512) void k(float *out, int n, int kindA, int kindB, int only)
{
    __shared__ __attribute__((aligned(16))) float lds[8][1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0;
    const int kind = wave < 4 ? kindA : kindB;
    if (only == 0 || (only == 1 && wave < 4) || (only == 2 && wave >= 4)) work(kind, n, acc, lds[wave], lane);
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

float run(float *d, int blocks, int n, int a, int b, int only)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, 4, a, b, only);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, n, a, b, only);
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
    const char *names[6] = {"ds_read_b32", "v_add_f32", "mfma_f32", "ds_write_b128", "ds_add_u32(random)", "mfma_bf16_16x16x32"};
    const int n = 3000;
    const int pairs[][2] = {{0, 1}, {0, 2}, {3, 1}, {3, 2}, {4, 1}, {4, 2}, {0, 3}, {5, 1}, {0, 5}, {4, 5}, {5, 2}};
    for (auto &pr : pairs) {
        const float ta = run(d, ncu, n, pr[0], pr[1], 1), tb = run(d, ncu, n, pr[0], pr[1], 2), tab = run(d, ncu, n, pr[0], pr[1], 0);
        printf("%-20s alone %.3f ms | %-12s alone %.3f ms | together %.3f ms  (max %.3f, sum %.3f)\n",
               names[pr[0]], ta, names[pr[1]], tb, tab, ta > tb ? ta : tb, ta + tb);
    }
    return 0;
}
