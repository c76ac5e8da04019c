// Per-track feature preparation on the device: the pooling every algorithm
// applies to the raw (T0, 12) chroma of a track before the pairwise stage.
//   Serra09 : rqa_serra09.py:44-53  -- librosa.util.sync(chroma.T, arange(0, T0, fac), aggregate=np.median).T
//   SiMPle  : simple_silva.py:34-43, 56-66 -- WIN / SKIP window means, Hann smoothing, L2 per frame
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acx {

constexpr int POOL_MAXFAC = 64;    // frames per median block supported (reference default: 40)
constexpr int POOL_FPB = 16;       // pooled frames per workgroup

// track of pooled frame p: the t with poff[t] <= p < poff[t + 1] (tracks without frames are skipped)
__device__ __forceinline__ int track_of(const int64_t *__restrict__ poff, int n_tracks, int64_t p)
{
    int lo = 0, hi = n_tracks;                       // poff[lo] <= p < poff[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (poff[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------
// P0: non-finite values.  One NaN / Inf frame poisons more than its own track: the band kernel's edge
// tiles read a neighbouring track's frames unclamped (the cells they feed lie outside the matrix and
// are +inf by construction -- unless the neighbour turns the sum into NaN, which the clamp at 0 makes a
// spurious smallest element).  So every pool is scanned once on its way in.  `x` holds `n` values of
// the rows [row_base, ...) of a packed (rows, dim) array whose track t owns rows off[t] .. off[t + 1].
// A NaN is replaced by 0 when zero_nan, an Inf when zero_inf (the reference itself zeroes NaN MFCCs,
// earlyfusion_traile.py:105); anything else non-finite is reported: res[0] = smallest offending track
// (atomicMin; initialised to INT_MAX), res[1] += values zeroed.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nonfinite_kernel(T *__restrict__ x, int64_t n, int dim, int64_t row_base,
                                                        const int64_t *__restrict__ off, int n_tracks, int zero_nan,
                                                        int zero_inf, int *__restrict__ res)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    int zeroed = 0;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        const T v = x[k];
        const bool is_nan = !(v == v);
        const bool is_inf = !is_nan && !(v - v == (T)0);
        if (!is_nan && !is_inf) continue;
        if ((is_nan && zero_nan) || (is_inf && zero_inf)) { x[k] = (T)0; ++zeroed; continue; }
        atomicMin(&res[0], off ? track_of(off, n_tracks, row_base + k / dim) : (int)(row_base + k / dim));     // no offsets: one row per track
    }
    if (zeroed) atomicAdd(&res[1], zeroed);
}

// ------------------------------------------------------------------------------------
// P1: block medians.  Workgroup = 16 pooled frames; the raw block of a pooled frame (<= 64
// frames x 12 bins, contiguous) is staged in LDS; thread (frame, bin) finds the two middle order
// statistics by rank counting (ties ranked by position) and writes their f32 mean -- np.median
// of an even count is mean(two middle values) in the input dtype, of an odd count the middle.
// raw holds the frames [raw_base, ...) of the pool; roff / poff are the raw / pooled offsets.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_median_kernel(const float *__restrict__ raw, int64_t raw_base,
                                                          const int64_t *__restrict__ roff, const int64_t *__restrict__ poff,
                                                          int n_tracks, int64_t p_begin, int64_t p_end, int fac,
                                                          float *__restrict__ pooled)
{
    __shared__ float tile[POOL_FPB][POOL_MAXFAC * 12];
    __shared__ int64_t s_start[POOL_FPB];
    __shared__ int s_cnt[POOL_FPB];
    const int tid = threadIdx.x;
    const int64_t p0 = p_begin + (int64_t)blockIdx.x * POOL_FPB;
    if (tid < POOL_FPB) {
        const int64_t p = p0 + tid;
        int cnt = 0;
        int64_t start = 0;
        if (p < p_end) {
            const int t = track_of(poff, n_tracks, p);
            start = roff[t] + (p - poff[t]) * fac;
            const int64_t left = roff[t + 1] - start;
            cnt = (int)(left < fac ? left : fac);
        }
        s_start[tid] = start;
        s_cnt[tid] = cnt;
    }
    __syncthreads();
    for (int f = 0; f < POOL_FPB; ++f) {
        const int n = s_cnt[f] * 12;
        const float *src = raw + (s_start[f] - raw_base) * 12;
        for (int k = tid; k < n; k += 256) tile[f][k] = src[k];
    }
    __syncthreads();
    if (tid < POOL_FPB * 12) {
        const int f = tid / 12, b = tid - 12 * f;
        const int cnt = s_cnt[f];
        if (cnt > 0) {
            const int k1 = (cnt - 1) >> 1, k2 = cnt >> 1;
            float lo = 0.0f, hi = 0.0f;
            for (int a = 0; a < cnt; ++a) {
                const float va = tile[f][a * 12 + b];
                int rank = 0;
                for (int q = 0; q < cnt; ++q) {
                    const float vq = tile[f][q * 12 + b];
                    rank += (vq < va || (vq == va && q < a)) ? 1 : 0;
                }
                if (rank == k1) lo = va;
                if (rank == k2) hi = va;
            }
            pooled[(p0 + f - p_begin) * 12 + b] = (k1 == k2) ? lo : (lo + hi) / 2.0f;
        }
    }
}

// ------------------------------------------------------------------------------------
// P2: SiMPle features of one track per workgroup.  Stage 1: pooled[i][c] = mean of raw frames
// [i skip, i skip + win) clipped to the track (f32, frames added one after the other in time
// order, then divided by the count -- numpy's reduction order for this strided axis), kept in
// LDS as f64.  Stage 2: 'same' convolution along time with the normalised Hann window (nw taps,
// zero fill), then every frame divided by its L2 norm over the 12 bins (norm < tiny: unscaled).
// ------------------------------------------------------------------------------------
constexpr int SIMPLE_PREP_CHUNK = 480;     // output frames per pass; + 2 x 16 frames of halo = 512 pooled frames in LDS
constexpr int SIMPLE_PREP_MAXW = 16;

struct SmoothWin { double w[SIMPLE_PREP_MAXW]; int nw; };

__global__ __launch_bounds__(256) void simple_prep_kernel(const float *__restrict__ raw, int64_t raw_base,
                                                          const int64_t *__restrict__ roff, const int64_t *__restrict__ poff,
                                                          int t_begin, int win, int skip, SmoothWin sw,
                                                          double *__restrict__ feats, int64_t p_base)
{
    __shared__ double pooled[(SIMPLE_PREP_CHUNK + 2 * SIMPLE_PREP_MAXW) * 12];
    const int t = t_begin + blockIdx.x;
    const int64_t r0 = roff[t], T0 = roff[t + 1] - r0;
    const int n = (int)(poff[t + 1] - poff[t]);
    const float *x = raw + (r0 - raw_base) * 12;
    const int off = (sw.nw - 1) / 2;
    double *out = feats + (poff[t] - p_base) * 12;
    // a track of any length: SIMPLE_PREP_CHUNK output frames per pass, their pooled neighbours as halo
    for (int i0 = 0; i0 < n; i0 += SIMPLE_PREP_CHUNK) {
        const int lo = i0 - SIMPLE_PREP_MAXW < 0 ? 0 : i0 - SIMPLE_PREP_MAXW;
        int hi = i0 + SIMPLE_PREP_CHUNK + SIMPLE_PREP_MAXW;
        hi = hi > n ? n : hi;
        for (int k = threadIdx.x; k < (hi - lo) * 12; k += 256) {
            const int i = lo + k / 12, c = k % 12;
            const int64_t a = (int64_t)i * skip;
            int64_t e = a + win;
            if (e > T0) e = T0;
            float acc = 0.0f;
            for (int64_t f = a; f < e; ++f) acc = acc + x[f * 12 + c];
            pooled[k] = (double)(acc / (float)(e - a));
        }
        __syncthreads();
        const int i1 = i0 + SIMPLE_PREP_CHUNK > n ? n : i0 + SIMPLE_PREP_CHUNK;
        for (int i = i0 + threadIdx.x; i < i1; i += 256) {
            double v[12];
            double ss = 0.0;
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                double acc = 0.0;
                for (int k = 0; k < sw.nw; ++k) {
                    const int src = i + off - k;
                    if (src >= 0 && src < n) acc = acc + pooled[(src - lo) * 12 + c] * sw.w[k];
                }
                v[c] = acc;
                ss = ss + acc * acc;
            }
            double nrm = __builtin_sqrt(ss);
            if (nrm < 2.2250738585072014e-308) nrm = 1.0;
#pragma unroll
            for (int c = 0; c < 12; ++c) out[i * 12 + c] = v[c] / nrm;
        }
        __syncthreads();
    }
}

}  // namespace acx
