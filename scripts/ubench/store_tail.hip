// How fast can ONE workgroup of 8 waves put a 256 x 128 f32 output tile (128 KB) into HBM-backed memory -- the epilogue of
// ef_gemm_rect_bf16x3_kernel -- and what does the shape of a store instruction cost?  (profiles/r05_ef.md: the epilogue is 14.6 % of
// a GEMM tile until its last store is ISSUED, + 3 % until acknowledged; MI355X_MICROARCH: "store-issue-bound, ~7 B / cycle / CU".)
// Every wave owns 64 x 64 cells of the tile and writes them with 16 global_store_dwordx4 (1 KB each) in one of the shapes:
//   A  8 rows x 128 B   (the kernel's: two neighbouring 16 x 16 sub-tiles turned through LDS, 8 full lines per instruction)
//   C  16 rows x 64 B   (round 3's first version: a lane's four accumulator columns, 16 half lines)
//   D  4 rows x 256 B   (four sub-tiles turned together)
//   B  1 KB contiguous  (a tiled matrix layout: what the statistics kernels would then have to read)
// with plain or non-temporal stores, at W workgroups in flight (one per CU: 256; fewer: the rest of the chip idle), R tiles each.
// Prints cycles per tile from the barrier to the last store issued / acknowledged (mean over waves) and bytes per cycle and CU.
//   hipcc --offload-arch=gfx950 -O3 -w -o build_ab/store_tail scripts/ubench/store_tail.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool NT>
__global__ __launch_bounds__(512) void tail(float *out, long pitch, int tiles_x, int rounds, unsigned long long *clk, int between)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = f32x4{(float)lane, (float)i, (float)wave, 1.0f};
    unsigned long long issued = 0, acked = 0;
    for (int r = 0; r < rounds; ++r) {
        const long tile = (long)r * gridDim.x + blockIdx.x;
        const long ty = tile / tiles_x, tx = tile % tiles_x;
        float *base = out + (ty * 256 + 64 * wr) * pitch + tx * 128 + 64 * wc;
        // something between the tiles, so that the stores of one tile have drained before the next (the k loop's stand-in)
        for (int i = 0; i < between; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = v[k] * 1.0001f + 0.5f;
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float *p;
            if (SHAPE == 0) {            // A: unit u = i / 2 (a = u / 2, column half u % 2), rows tr / 8 + tr
                const int u = i >> 1, a = u >> 1, bp = u & 1, tr = lane >> 3, tc = 4 * (lane & 7);
                p = base + (long)(16 * a + 8 * (i & 1) + tr) * pitch + 32 * bp + tc;
            } else if (SHAPE == 1) {     // C: sub-tile (a, b) = (i / 4, i % 4), row lane % 16, columns 4 (lane / 16)
                p = base + (long)(16 * (i >> 2) + (lane & 15)) * pitch + 16 * (i & 3) + 4 * (lane >> 4);
            } else if (SHAPE == 2) {     // D: rows 4 i + lane / 16, columns 4 (lane % 16)
                p = base + (long)(4 * i + (lane >> 4)) * pitch + 4 * (lane & 15);
            } else {                     // B: tiled layout, 1 KB per instruction
                p = out + tile * 32768 + wave * 4096 + i * 256 + 4 * lane;
            }
            if (NT) __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4 *>(p));
            else *reinterpret_cast<f32x4 *>(p) = v[i];
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t2 = __builtin_readcyclecounter();
        issued += t1 - t0;
        acked += t2 - t0;
    }
    if (lane == 0) { atomicAdd(&clk[0], issued); atomicAdd(&clk[1], acked); atomicAdd(&clk[2], (unsigned long long)rounds); }
}

template <int SHAPE, bool NT>
void run(const char *label, float *d, long pitch, int tiles_x, int wgs, int rounds, int between, unsigned long long *dclk)
{
    unsigned long long z[3] = {0, 0, 0}, h[3];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((tail<SHAPE, NT>), dim3(wgs), dim3(512), 0, 0, d, pitch, tiles_x, rounds, dclk, between);
    hipDeviceSynchronize();
    hipMemcpy(dclk, z, sizeof(z), hipMemcpyHostToDevice);
    hipEventRecord(e0);
    hipLaunchKernelGGL((tail<SHAPE, NT>), dim3(wgs), dim3(512), 0, 0, d, pitch, tiles_x, rounds, dclk, between);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost);
    const double per_issue = (double)h[0] / (double)h[2], per_ack = (double)h[1] / (double)h[2];
    printf("%-34s %3d workgroups x %3d tiles, %5d filler: issued %7.0f ticks / tile, acknowledged %7.0f; launch %.3f ms = %.2f TB/s\n", label, wgs, rounds,
           between, per_issue, per_ack, ms, (double)wgs * rounds * 131072.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    const int tiles_x = 32, max_tiles = 256 * 64;
    const long pitch = 128L * tiles_x + 64;                      // (not a power of two: rows land on different channels)
    float *d;
    unsigned long long *dclk;
    const size_t bytes = (size_t)(max_tiles / tiles_x + 1) * 256 * pitch * 4;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&dclk, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, bytes);
    // tick length: s_memtime against a known kernel duration is not needed -- ratios between shapes are what matters; the launch times give absolute rates
    for (int between : {0, 400}) {
        for (int wgs : {256, 32}) {
            const int rounds = 32;
            run<0, true>("A  8 rows x 128 B, non-temporal", d, pitch, tiles_x, wgs, rounds, between, dclk);
            run<0, false>("A  8 rows x 128 B, plain", d, pitch, tiles_x, wgs, rounds, between, dclk);
            run<1, true>("C  16 rows x 64 B, non-temporal", d, pitch, tiles_x, wgs, rounds, between, dclk);
            run<2, true>("D  4 rows x 256 B, non-temporal", d, pitch, tiles_x, wgs, rounds, between, dclk);
            run<2, false>("D  4 rows x 256 B, plain", d, pitch, tiles_x, wgs, rounds, between, dclk);
            run<3, true>("B  1 KB contiguous, non-temporal", d, pitch, tiles_x, wgs, rounds, between, dclk);
            run<3, false>("B  1 KB contiguous, plain", d, pitch, tiles_x, wgs, rounds, between, dclk);
        }
    }
    return 0;
}
