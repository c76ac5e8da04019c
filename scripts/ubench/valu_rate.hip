// Micro-benchmark (development aid, not product): issue rate of the VALU / LDS instruction
// classes the band kernel leans on, on gfx950.  Prints cycles per wave-instruction per SIMD
// with W waves resident per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 64
#define ITER 256

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int n, unsigned seed)
{
    __shared__ unsigned hist[4][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    float b = 1.0001f, c = 0.5f;
    unsigned u0 = lane, u1 = lane * 3, u2 = lane * 5, u3 = lane * 7;
    unsigned addr = ((lane * 2654435761u + seed) >> 21) * 4;   // random slot 0..2047
    for (int i = threadIdx.x; i < 4 * 2048; i += 256) (&hist[0][0])[i] = 0;
    __syncthreads();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if constexpr (OP == 0) {
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 1) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if constexpr (OP == 2) {
                asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                             "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));
            } else if constexpr (OP == 3) {
                asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 4) {   // cmp + cndmask pairs
                asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            } else if constexpr (OP == 5) {   // packed add: 8 instr = 16 adds
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(double *)&b));
            } else if constexpr (OP == 6) {   // LDS atomic add, random slots (no return)
                unsigned *h = &hist[wave][0];
                unsigned la = (unsigned)(size_t)h + addr;
                asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:4\n ds_add_u32 %0, %1 offset:8\n ds_add_u32 %0, %1 offset:12\n"
                             "ds_add_u32 %0, %1 offset:16\n ds_add_u32 %0, %1 offset:20\n ds_add_u32 %0, %1 offset:24\n ds_add_u32 %0, %1 offset:28\n"
                             :: "v"(la), "v"(u1) : "memory");
                addr = (addr * 5 + 4 * 77) & (2047 * 4 - 28 > 0 ? 8188 : 0);
                addr = addr > 8160 ? 8160 : addr;
            } else if constexpr (OP == 7) {   // DPP add (row_shr:1)
                asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (OP == 8) {   // v_min3 / v_max3
                asm volatile("v_min3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_min3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n"
                             "v_min3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_min3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 9) {   // cvt + and_or + lshl_add (int ops)
                asm volatile("v_cvt_u32_f32 %0, %4\n v_and_or_b32 %1, %0, %2, %3\n v_lshl_add_u32 %2, %1, 2, %3\n v_add3_u32 %3, %0, %1, %2\n"
                             "v_cvt_u32_f32 %0, %5\n v_and_or_b32 %1, %0, %2, %3\n v_lshl_add_u32 %2, %1, 2, %3\n v_add3_u32 %3, %0, %1, %2\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));
            } else if constexpr (OP == 12) {  // transcendental: exp2
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (OP == 13) {  // transcendental: reciprocal
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (OP == 14) {  // v_mul_f32
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 10) {  // ds_read_b32 conflict-free
                unsigned la = (unsigned)(size_t)&hist[wave][0] + lane * 4;
                unsigned t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                             "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"
                             : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3), "=v"(t4), "=v"(t5), "=v"(t6), "=v"(t7) : "v"(la) : "memory");
                u0 += t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
            } else if constexpr (OP == 11) {  // ds_write_b32 conflict-free
                unsigned la = (unsigned)(size_t)&hist[wave][0] + lane * 4;
                asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n"
                             "ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %1 offset:1280\n ds_write_b32 %0, %1 offset:1536\n ds_write_b32 %0, %1 offset:1792\n"
                             :: "v"(la), "v"(u1) : "memory");
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3) + (float)hist[wave][lane] + (float)addr;
}

template <int OP>
void run(const char *name, float *d, int wg_per_cu, int ncu)
{
    const int n = ITER;
    const int blocks = wg_per_cu * ncu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 8, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, n, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // each wave issues n*REP instrs; waves per SIMD = wg_per_cu (256 threads = 4 waves = 1 per SIMD)
    const double instr_per_simd = (double)n * REP * wg_per_cu;
    const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cycles@2.4GHz per wave-instr per SIMD\n", name, wg_per_cu, ms, cyc);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
    float *d; hipMalloc(&d, sizeof(float) * 256 * 8 * ncu);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_f32", d, w, ncu);
        run<1>("v_fma_f32", d, w, ncu);
        run<2>("v_add_u32", d, w, ncu);
        run<3>("v_max_f32", d, w, ncu);
        run<4>("v_cmp+v_cndmask", d, w, ncu);
        run<5>("v_pk_add_f32 (2 adds/instr)", d, w, ncu);
        run<7>("v_add_f32_dpp", d, w, ncu);
        run<8>("v_min3/max3_f32", d, w, ncu);
        run<9>("cvt/and_or/lshl_add/add3", d, w, ncu);
        run<14>("v_mul_f32", d, w, ncu);
        run<12>("v_exp_f32", d, w, ncu);
        run<13>("v_rcp_f32", d, w, ncu);
        run<6>("ds_add_u32 random", d, w, ncu);
        run<10>("ds_read_b32", d, w, ncu);
        run<11>("ds_write_b32", d, w, ncu);
    }
    return 0;
}
