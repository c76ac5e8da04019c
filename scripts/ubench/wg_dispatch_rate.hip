// How fast does one MI355X hand out workgroups?  band2_kernel's short classes launch 0.8-1.4 M workgroups of 128 / 256 threads per pass,
// each alive for a few microseconds: if the dispatchers cannot start them faster than the CUs finish them, the launch is bound by the
// dispatch rate and not by the SIMDs (profiles/r05_narrow_classes.md).  Kernels here do `work` dependent FMAs per lane and touch their
// LDS allocation once; grid = (bands, pairs) like the real launch.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o wg_dispatch_rate wg_dispatch_rate.hip && ./wg_dispatch_rate
#include <hip/hip_runtime.h>

#include <cstdio>

template <int THREADS, int LDS_FLOATS>
__global__ __launch_bounds__(THREADS) void k(float *out, int work)
{
    __shared__ float s[LDS_FLOATS];
    float x = (float)threadIdx.x;
    for (int i = 0; i < work; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    s[threadIdx.x] = x;
    __syncthreads();
    if (x == 1.2345e-30f) out[blockIdx.x] = s[(threadIdx.x + 1) % THREADS];
}

template <int THREADS, int LDS_FLOATS>
void run(const char *label, int bands, int pairs, int work, float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid(bands, pairs, 1);
    hipLaunchKernelGGL((k<THREADS, LDS_FLOATS>), grid, dim3(THREADS), 0, 0, d, work);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<THREADS, LDS_FLOATS>), grid, dim3(THREADS), 0, 0, d, work);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double wgs = (double)bands * pairs;
    printf("%-34s work %5d  grid %4d x %6d = %8.0f workgroups  %7.3f ms  %7.1f workgroups/us  %7.1f waves/us\n", label, work, bands, pairs, wgs, best,
           wgs / best / 1e3, wgs * (THREADS / 64) / best / 1e3);
}

int main()
{
    float *d;
    hipMalloc(&d, 1 << 20);
    const int works[] = {0, 256, 1024, 2048, 4096};
    for (int w : works) {
        run<128, 2752>("128 threads, 10.75 KB LDS", 18, 44850, w, d);
        run<256, 5504>("256 threads, 21.5 KB LDS", 18, 44850, w, d);
        run<256, 2752>("256 threads, 10.75 KB LDS", 18, 44850, w, d);
        run<512, 11008>("512 threads, 43 KB LDS", 18, 44850, w, d);
        run<64, 1376>("64 threads, 5.4 KB LDS", 18, 44850, w, d);
    }
    // Where do the waves of a SMALL launch go?  One-wave workgroups with a long dependent chain: if the time does not grow until the
    // grid exceeds 1024 (one wave per SIMD) they are spread over the chip; if it is already several times the one-wave time at a few
    // hundred, the dispatcher packs them onto few CUs and a latency-bound kernel (the alignment sweeps) becomes issue-bound there.
    printf("\nplacement of one-wave workgroups (20000 dependent FMAs each):\n");
    const int grids[] = {1, 64, 256, 512, 752, 1024, 2048, 4096, 8192};
    for (int g : grids) run<64, 64>("64 threads, 256 B LDS", g, 1, 20000, d);
    for (int g : grids) run<256, 64>("256 threads, 256 B LDS", (g + 3) / 4, 1, 20000, d);
    return 0;
}
