// Micro-benchmark (development aid): what does v_cndmask_b32 cost when SEVERAL selects read one comparison's mask?
// scripts/ubench/valu_table measured 22.9 cycles per v_cndmask_b32 ... vcc in a stream of nothing else (4.2 with the mask in an SGPR
// pair, 3.95 per instruction in v_cmp / v_cndmask pairs).  Patterns here, eight waves per SIMD, cycles per instruction and SIMD:
//   A: v_cmp vcc + N selects on vcc            B: v_cmp s[22:23] + N selects on s[22:23]          (N = 1, 2, 4, 7)
//   C: selects on vcc with an independent v_add_f32 between them
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
template <int PAT, int N>
__global__ __launch_bounds__(256) void k(float *out, int n)
{
    const int lane = threadIdx.x & 63;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    float b = 1.0001f, c = 0.5f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if constexpr (PAT == 0) {
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
                if constexpr (N >= 2) asm volatile("v_cndmask_b32 %2, %2, %9, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
                if constexpr (N >= 4) asm volatile("v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
                if constexpr (N >= 7) asm volatile("v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            } else if constexpr (PAT == 1) {
                asm volatile("v_cmp_lt_f32 s[22:23], %0, %8\n v_cndmask_b32 %1, %1, %9, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s22", "s23");
                if constexpr (N >= 2) asm volatile("v_cndmask_b32 %2, %2, %9, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s22", "s23");
                if constexpr (N >= 4) asm volatile("v_cndmask_b32 %3, %3, %9, s[22:23]\n v_cndmask_b32 %4, %4, %9, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s22", "s23");
                if constexpr (N >= 7) asm volatile("v_cndmask_b32 %5, %5, %9, s[22:23]\n v_cndmask_b32 %6, %6, %9, s[22:23]\n v_cndmask_b32 %7, %7, %9, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s22", "s23");
            } else if constexpr (PAT == 3) {
                // D: the recurrence bitmap's triple (band2_kernel): v_min_f32, v_cmp_le_f32 vcc, v_addc_co_u32 acc, vcc, acc, acc, vcc -- four of them
                asm volatile("v_min_f32 %4, %0, %8\n v_cmp_le_f32 vcc, %1, %4\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n"
                             "v_min_f32 %4, %1, %8\n v_cmp_le_f32 vcc, %2, %4\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n"
                             "v_min_f32 %4, %2, %8\n v_cmp_le_f32 vcc, %3, %4\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n"
                             "v_min_f32 %4, %3, %8\n v_cmp_le_f32 vcc, %0, %4\n v_addc_co_u32 %5, vcc, %5, %5, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            } else if constexpr (PAT == 4) {
                // E: selects of CONSTANTS on vcc (the VOP3 form hipcc uses for `cond ? 1 : 0`), four adjacent
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e64 %1, 0, 1, vcc\n v_cndmask_b32_e64 %2, 0, 1, vcc\n v_cndmask_b32_e64 %3, 0, 1, vcc\n v_cndmask_b32_e64 %4, 0, 1, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            } else if constexpr (PAT == 5) {
                // F: four comparisons into vcc, each with ONE select behind it (no adjacency)
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            } else if constexpr (PAT == 6) {
                // G: a DPP reduction chain as the selections run them: four DEPENDENT v_min_u32_dpp on one register
                asm volatile("v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                             "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n s_nop 1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            } else {
                // C: one comparison, then select / add / select / add ... (N selects, N adds)
                asm volatile("v_cmp_lt_f32 vcc, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
                asm volatile("v_cndmask_b32 %1, %1, %9, vcc\n v_add_f32 %5, %5, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_add_f32 %6, %6, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_add_f32 %7, %7, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_add_f32 %0, %0, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int PAT, int N>
void run(float *d, int ncu, const char *name)
{
    const int w = 8, per = PAT == 2 ? 9 : (PAT == 3 ? 12 : (PAT == 4 ? 5 : (PAT == 5 ? 8 : (PAT == 6 ? 8 : 1 + N))));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PAT, N>), dim3(w * ncu), dim3(256), 0, 0, d, 8);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<PAT, N>), dim3(w * ncu), dim3(256), 0, 0, d, ITER);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * 8 * w);
    printf("%-58s %7.2f cycles per group of %d = %.2f per instruction\n", name, cyc, per, cyc / per);
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float *d; (void)hipMalloc(&d, sizeof(float) * 256 * 8 * p.multiProcessorCount);
    const int ncu = p.multiProcessorCount;
    run<0, 1>(d, ncu, "v_cmp vcc + 1 select on vcc");
    run<0, 2>(d, ncu, "v_cmp vcc + 2 selects on vcc");
    run<0, 4>(d, ncu, "v_cmp vcc + 4 selects on vcc");
    run<0, 7>(d, ncu, "v_cmp vcc + 7 selects on vcc");
    run<1, 1>(d, ncu, "v_cmp s[22:23] + 1 select on it");
    run<1, 2>(d, ncu, "v_cmp s[22:23] + 2 selects on it");
    run<1, 4>(d, ncu, "v_cmp s[22:23] + 4 selects on it");
    run<1, 7>(d, ncu, "v_cmp s[22:23] + 7 selects on it");
    run<2, 4>(d, ncu, "v_cmp vcc + 4 x (select on vcc, v_add_f32)");
    run<3, 4>(d, ncu, "4 x (v_min_f32, v_cmp_le_f32 vcc, v_addc_co_u32 on vcc)");
    run<4, 4>(d, ncu, "v_cmp vcc + 4 x v_cndmask_b32_e64 v, 0, 1, vcc");
    run<5, 4>(d, ncu, "4 x (v_cmp vcc, one select on vcc)");
    run<6, 4>(d, ncu, "4 dependent v_min_u32_dpp, s_nop 1 behind each");
    return 0;
}
