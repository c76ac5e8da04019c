// Micro-benchmark (development aid, not product): HBM write / read rates in the access patterns of the
// stored-distance pipeline.   hipcc --offload-arch=gfx950 -O3 -o hbm_rw hbm_rw.hip
//   fill:   each wave writes one 8 KB row with 8 x 16-byte stores per lane (band_kernel's Z rows)
//   stream: each lane reads 8 x 16 bytes, coalesced (ideal read)
//   gather: band_read_kernel's pattern -- thread = row j, 32 bytes at column offset 8 b + 7 - (j & 7)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void fill(float4 *z, long rows)
{
    const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 *p = z + row * 512 + (threadIdx.x & 63) * 8;
    const float v = (float)row;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = make_float4(v, v + 1, v + 2, v + 3);
}
__global__ __launch_bounds__(512) void fill_coalesced(float4 *z, long rows)
{
    const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 *p = z + row * 512 + (threadIdx.x & 63);
    const float v = (float)row;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[64 * j] = make_float4(v, v + 1, v + 2, v + 3);
}
__global__ __launch_bounds__(512) void stream(const float4 *z, long rows, float *out)
{
    const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 *p = z + row * 512 + (threadIdx.x & 63);
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float4 v = p[64 * j]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) out[0] = s;
}
// one "pair" = 2048 rows x 2048 floats; workgroup = (band, pair); XCD-aware band order as in band_read_kernel
typedef float f32x4n __attribute__((ext_vector_type(4), aligned(4)));
template <bool XCD>
__global__ __launch_bounds__(512) void gather(const float *z, float *out)
{
    const int bx = blockIdx.x;
    const int band = XCD ? (bx & 7) * (int)(gridDim.x >> 3) + (bx >> 3) : bx;
    const float *zp = z + (size_t)blockIdx.y * 2048 * 2048 + 8 * band + 7 - (threadIdx.x & 7);
    float s = 0;
    f32x4n v0[4], v1[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int j = threadIdx.x + 512 * n;
        const f32x4n *src = (const f32x4n *)(zp + (size_t)j * 2048);
        v0[n] = src[0]; v1[n] = src[1];
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) s += v0[n].x + v0[n].y + v0[n].z + v0[n].w + v1[n].x + v1[n].y + v1[n].z + v1[n].w;
    if (s == 12345.f) out[0] = s;
}

// as gather<true>, but a PAIR of neighbouring lanes fetches the two 16-byte halves of one row's 32 bytes in the same
// instruction (one L2 request instead of two); a thread covers 8 rows with 8 loads
__global__ __launch_bounds__(512) void gather_pairs(const float *z, float *out)
{
    const int bx = blockIdx.x;
    const int band = (bx & 7) * (int)(gridDim.x >> 3) + (bx >> 3);
    const int t = threadIdx.x, jr = t >> 1, h = t & 1;
    float s = 0;
    f32x4n v[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int j = jr + 256 * n;
        v[n] = *(const f32x4n *)(z + (size_t)blockIdx.y * 2048 * 2048 + (size_t)j * 2048 + 8 * band + 7 - (j & 7) + 4 * h);
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) s += v[n].x + v[n].y + v[n].z + v[n].w;
    if (s == 12345.f) out[0] = s;
}

// transposed store of the column pass: workgroup = 8 rows j0 .. j0 + 7, a PAIR of lanes writes the 32 bytes
// Zt[c][j0 .. j0 + 7] of query c (8 queries per thread)
template <bool XCD>
__global__ __launch_bounds__(512) void scatter_pairs(float *z)
{
    const int bx = blockIdx.x;
    const int band = XCD ? (bx & 7) * (int)(gridDim.x >> 3) + (bx >> 3) : bx;
    const int t = threadIdx.x, cr = t >> 1, h = t & 1;
    const float v = (float)t;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int c = cr + 256 * n;
        *(float4 *)(z + (size_t)blockIdx.y * 2048 * 2048 + (size_t)c * 2048 + 8 * band + 4 * h) = make_float4(v, v + 1, v + 2, v + 3);
    }
}

// chunked layout: Zc[pc][j][CW + 8] (CW positions per chunk + the first 8 of the next one): the records of
// consecutive rows j are adjacent, so one gather instruction sweeps a contiguous region
template <int CW>
__global__ __launch_bounds__(512) void gather_chunk(const float *z, float *out)
{
    const int bx = blockIdx.x;
    const int band = (bx & 7) * (int)(gridDim.x >> 3) + (bx >> 3);
    constexpr int REC = CW + 8, BPC = CW / 8;
    const float *zp = z + (size_t)blockIdx.y * (2048 / CW) * 2048 * REC + (size_t)(band / BPC) * 2048 * REC + 8 * (band % BPC) + 7 - (threadIdx.x & 7);
    float s = 0;
    f32x4n v0[4], v1[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int j = threadIdx.x + 512 * n;
        const f32x4n *src = (const f32x4n *)(zp + (size_t)j * REC);
        v0[n] = src[0]; v1[n] = src[1];
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) s += v0[n].x + v0[n].y + v0[n].z + v0[n].w + v1[n].x + v1[n].y + v1[n].z + v1[n].w;
    if (s == 12345.f) out[0] = s;
}

int main()
{
    const long pairs = 1024;                       // 16 GiB
    const long rows = pairs * 2048;
    float4 *z; float *out;
    CK(hipMalloc(&z, rows * 8192 * 5 / 4 + 4096)); CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fill, dim3(rows / 8), dim3(512), 0, 0, z, rows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("fill    %.2f ms  %.2f TB/s\n", ms, rows * 8192.0 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fill_coalesced, dim3(rows / 8), dim3(512), 0, 0, z, rows);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("fill (1 KB per instruction) %.2f ms  %.2f TB/s\n", ms, rows * 8192.0 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream, dim3(rows / 8), dim3(512), 0, 0, z, rows, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream  %.2f ms  %.2f TB/s\n", ms, rows * 8192.0 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather<true>, dim3(248, pairs), dim3(512), 0, 0, (const float *)z, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("gather (XCD-aware bands)  %.2f ms  %.2f TB/s useful\n", ms, pairs * 248.0 * 2048 * 32 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_pairs, dim3(248, pairs), dim3(512), 0, 0, (const float *)z, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("gather, lane pairs per row  %.2f ms  %.2f TB/s useful\n", ms, pairs * 248.0 * 2048 * 32 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(scatter_pairs<true>, dim3(256, pairs), dim3(512), 0, 0, (float *)z);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("scatter 32-byte pieces, XCD-aware  %.2f ms  %.2f TB/s\n", ms, pairs * 256.0 * 2048 * 32 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(scatter_pairs<false>, dim3(256, pairs), dim3(512), 0, 0, (float *)z);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("scatter 32-byte pieces, plain order  %.2f ms  %.2f TB/s\n", ms, pairs * 256.0 * 2048 * 32 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_chunk<32>, dim3(256, pairs), dim3(512), 0, 0, (const float *)z, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("gather, chunks of 32 + 8 positions  %.2f ms  %.2f TB/s useful\n", ms, pairs * 256.0 * 2048 * 32 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_chunk<64>, dim3(256, pairs), dim3(512), 0, 0, (const float *)z, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("gather, chunks of 64 + 8 positions  %.2f ms  %.2f TB/s useful\n", ms, pairs * 256.0 * 2048 * 32 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_chunk<128>, dim3(256, pairs), dim3(512), 0, 0, (const float *)z, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("gather, chunks of 128 + 8 positions  %.2f ms  %.2f TB/s useful\n", ms, pairs * 256.0 * 2048 * 32 / ms / 1e9);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather<false>, dim3(248, pairs), dim3(512), 0, 0, (const float *)z, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("gather (plain band order) %.2f ms  %.2f TB/s useful\n", ms, pairs * 248.0 * 2048 * 32 / ms / 1e9);
    }
    return 0;
}
