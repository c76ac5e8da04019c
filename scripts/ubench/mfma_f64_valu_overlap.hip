// Micro-benchmark (development aid): does v_mfma_f64_16x16x4_f64 of one wave run BESIDE the VALU work of another wave on the same SIMD?
// (SiMPle's 12-bin frame Gram on the f64 matrix pipe, the window sum as a sliding diagonal sum.  On MI355X the f64
// matrix peak equals the f64 vector peak -- 78.6 TFLOP/s --, so the Gram costs the same pipe time either way: the move pays only if
// the matrix instructions overlap with what stays on the vector pipe -- the f64 adds / minimum, the DPP and select moves of
// simple_kernel's sweep.)  Workgroups of 512 threads = 2 waves per SIMD.  mode 0: all waves MFMA; 1: all waves VALU; 2: first four
// waves MFMA, last four VALU (one of each per SIMD).   VOP 0: v_fma_f64, 1: v_add_f64, 2: v_add_f32, 3: v_mov_b32 dpp (wave_shr)
//   hipcc --offload-arch=gfx950 -O3 -w -o build_ab/mfma_f64_valu_overlap scripts/ubench/mfma_f64_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int VOP>
__global__ __launch_bounds__(512) void k(double *out, int n, int mode)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = lane + 4, x5 = lane + 5, x6 = lane + 6, x7 = lane + 7;
    float y0 = lane, y1 = lane + 1, y2 = lane + 2, y3 = lane + 3, y4 = lane + 4, y5 = lane + 5, y6 = lane + 6, y7 = lane + 7;
    double b = 1.0001;
    float bf = 1.0001f;
    if (do_mfma) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, b, a3, 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {     // 128 VALU per iteration beside 16 MFMA
                if (VOP == 0)
                    asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                                 "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
                else if (VOP == 1)
                    asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                                 "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
                else if (VOP == 2)
                    asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                                 "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                                 : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(bf));
                else
                    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1\n v_mov_b32_dpp %1, %2 wave_shr:1\n v_mov_b32_dpp %2, %3 wave_shr:1\n v_mov_b32_dpp %3, %4 wave_shr:1\n"
                                 "v_mov_b32_dpp %4, %5 wave_shr:1\n v_mov_b32_dpp %5, %6 wave_shr:1\n v_mov_b32_dpp %6, %7 wave_shr:1\n v_mov_b32_dpp %7, %0 wave_shr:1\n"
                                 : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}

template <int VOP>
float run(double *d, int blocks, int n, int mode)
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

template <int VOP>
void report(const char *name, double *d)
{
    const int blocks = 256, n = 2000;
    const float m = run<VOP>(d, blocks, n, 0), v = run<VOP>(d, blocks, n, 1), x = run<VOP>(d, blocks, n, 2);
    printf("%-22s all-MFMA %.3f ms (2 waves/SIMD x %d v_mfma_f64_16x16x4_f64)   all-VALU %.3f ms (2 waves/SIMD x %d)   1 MFMA wave + 1 VALU wave per SIMD %.3f ms\n"
           "   expected if they overlap fully: %.3f ms; if they serialize: %.3f ms\n", name, m, 16 * n, v, 128 * n, x, (m > v ? m : v) / 2, (m + v) / 2);
}

int main()
{
    double *d;
    (void)hipMalloc(&d, 256 * 512 * 8);
    report<0>("v_fma_f64", d);
    report<1>("v_add_f64", d);
    report<2>("v_add_f32", d);
    report<3>("v_mov_b32 dpp", d);
    return 0;
}
