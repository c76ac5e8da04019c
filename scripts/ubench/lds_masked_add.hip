// Micro-benchmark (development aid, not product): cost of an LDS atomic add when only a fraction of
// the lanes is active (exec-masked), random slots -- the filtered histogram of the band kernel's
// selection issues its atomics under such a mask.   hipcc --offload-arch=gfx950 -O3 -o lds_masked_add lds_masked_add.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 64
#define ITER 256

// ACTIVE = number of active lanes out of 64 (spread by a multiplicative hash)
template <int ACTIVE, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int n, unsigned seed)
{
    __shared__ unsigned hist[4][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = ((lane * 2654435761u + seed) >> 21) * 4;   // random slot 0..2047
    for (int i = threadIdx.x; i < 4 * 2048; i += 256) (&hist[0][0])[i] = 0;
    __syncthreads();
    const bool on = ((lane * 37 + 11) & 63) < ACTIVE;
    unsigned u1 = 1;
    if (on) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r) {
                unsigned la = (unsigned)(size_t)&hist[wave][0] + addr;
                if constexpr (MODE == 0) {
                    asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:4\n ds_add_u32 %0, %1 offset:8\n ds_add_u32 %0, %1 offset:12\n"
                                 "ds_add_u32 %0, %1 offset:16\n ds_add_u32 %0, %1 offset:20\n ds_add_u32 %0, %1 offset:24\n ds_add_u32 %0, %1 offset:28\n"
                                 :: "v"(la), "v"(u1) : "memory");
                } else {
                    asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:4\n ds_write_b32 %0, %1 offset:8\n ds_write_b32 %0, %1 offset:12\n"
                                 "ds_write_b32 %0, %1 offset:16\n ds_write_b32 %0, %1 offset:20\n ds_write_b32 %0, %1 offset:24\n ds_write_b32 %0, %1 offset:28\n"
                                 :: "v"(la), "v"(u1) : "memory");
                }
                addr = (addr * 5 + 4 * 77) & 8188;
                addr = addr > 8160 ? 8160 : addr;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)hist[wave][lane] + (float)addr;
}

template <int ACTIVE, int MODE>
void run(const char *name, float *d, int wg_per_cu, int ncu)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = wg_per_cu * ncu;
    hipLaunchKernelGGL((k<ACTIVE, MODE>), dim3(blocks), dim3(256), 0, 0, d, 8, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ACTIVE, MODE>), dim3(blocks), dim3(256), 0, 0, d, ITER, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)ITER * REP * wg_per_cu * 4;
    printf("%-14s active=%2d waves/SIMD=%d  %.3f ms  %.2f LDS cycles@2.4GHz per wave-instr (per CU)\n", name, ACTIVE, wg_per_cu, ms,
           ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.name, ncu);
    float *d; hipMalloc(&d, sizeof(float) * 256 * 8 * ncu);
    for (int w : {2, 4}) {
        run<64, 0>("ds_add_u32", d, w, ncu);
        run<32, 0>("ds_add_u32", d, w, ncu);
        run<16, 0>("ds_add_u32", d, w, ncu);
        run<12, 0>("ds_add_u32", d, w, ncu);
        run<8, 0>("ds_add_u32", d, w, ncu);
        run<4, 0>("ds_add_u32", d, w, ncu);
        run<1, 0>("ds_add_u32", d, w, ncu);
        run<64, 1>("ds_write_b32", d, w, ncu);
        run<16, 1>("ds_write_b32", d, w, ncu);
        run<8, 1>("ds_write_b32", d, w, ncu);
    }
    return 0;
}
