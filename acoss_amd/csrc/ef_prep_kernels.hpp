// EarlyFusion block features on the device (gfx950): what EarlyFusion.load_features builds per
// track on the CPU (reference acoss/algorithms/earlyfusion_traile.py:100-140 and resize_block,
// :214-247), for every block of every track in one launch.
//
//   block b of a track with beats o_0 < o_1 < ... (frame indices):
//     mfccs[b]   frames [o_b, o_{b+blocksize-1}) of the MFCCs, resized along time to
//                mfccs_per_block rows (anti-aliased), every coefficient made zero-mean over the
//                block, every row divided by its L2 norm (zero norm: unscaled); row-major
//     ssms[b]    Euclidean self-similarity of those rows, the cells (r, c) with c < r in row-major
//                order (the reference's D[I < J] with I, J = meshgrid(pix, pix))
//     chromas[b] frames [o_b, o_{b+blocksize}) of the chroma resized to chromas_per_block rows
//   chroma_med   per-bin median over ALL frames of the track
//
// The resize is skimage.transform.resize(x, (rows, d), anti_aliasing=True, mode='constant'): a Gaussian
// filter along time with sigma = max(0, (n / rows - 1) / 2), truncated at 4 sigma, zeros outside the
// block; linear interpolation on the pixel-centre grid in = (out + 0.5) n / rows - 0.5 with zeros
// outside; and skimage's output clip (clip=True): every value is clipped to [min, max] of the INPUT
// block -- over all its columns --, except that values equal to the fill value 0 stay 0 when 0 lies
// outside that range (it matters for all-positive chroma blocks, whose first / last rows blend with the
// zeros outside).  Pinned by tests/golden/efprep_skimage.npz: the reference's own load_features /
// resize_block run with scikit-image 0.18.3 (tests/golden/make_efprep_goldens.py).
// All arithmetic in f64 (the reference converts the block to float64), results stored as f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acx {

constexpr int EFP_MAXROWS = 64;      // rows per block after the resize (reference: 50 / 40)
constexpr int EFP_MAXDIM = 40;       // features per frame (13 MFCCs / 12 chroma bins)
constexpr int EFP_MAXRAD = 512;      // Gaussian radius (4 sigma): blocks of up to ~ 256 x rows frames

struct EfPrepParams {
    int blocksize;        // beats per block (20)
    int mfcc_rows;        // mfccs_per_block (50)
    int chroma_rows;      // chromas_per_block (40)
    int ncoef;            // MFCC coefficients per frame (13)
};

// One filtered sample: sum_k w[k] x[t + k - r][d] over the block's frames [0, n), zeros outside
__device__ __forceinline__ double efp_filtered(const float *__restrict__ x, int n, int dim, int d, int t,
                                               const double *w, int r)
{
    // (NaN samples count as 0: the reference zeroes them before anything else, earlyfusion_traile.py:108)
    auto sample = [&](int tt) { const float v = x[(size_t)tt * dim + d]; return v == v ? (double)v : 0.0; };
    if (r == 0) return (t >= 0 && t < n) ? sample(t) : 0.0;
    double acc = 0.0;
    int k0 = t - r < 0 ? r - t : 0, k1 = t + r > n - 1 ? n - 1 - t + r : 2 * r;
    for (int k = k0; k <= k1; ++k) acc += w[k] * sample(t + k - r);
    return acc;
}

// resize of frames [i1, i2) of X (rows x dim, f32) to `rows` rows into out[rows][dim] (LDS, f64)
__device__ __forceinline__ void efp_resize(const float *__restrict__ X, int64_t i1, int64_t i2, int dim, int rows,
                                           double *out, double *w, double *red, int tid, int nthreads)
{
    const int n = (int)(i2 - i1);
    const float *x = X + i1 * dim;
    // range of the input block (skimage's clip); NaN samples count as 0 like everywhere else
    {
        double mn = __builtin_inf(), mx = -__builtin_inf();
        for (int e = tid; e < n * dim; e += nthreads) {
            const float v0 = x[e];
            const double v = v0 == v0 ? (double)v0 : 0.0;
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
        }
        for (int off = 32; off >= 1; off >>= 1) {
            const double a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        __syncthreads();                                  // (red may still be read by the previous call's tail)
        if ((tid & 63) == 0) { red[2 * (tid >> 6)] = mn; red[2 * (tid >> 6) + 1] = mx; }
        __syncthreads();
    }
    double bmin = red[0], bmax = red[1];
    for (int k = 1; k < (nthreads >> 6); ++k) {
        bmin = red[2 * k] < bmin ? red[2 * k] : bmin;
        bmax = red[2 * k + 1] > bmax ? red[2 * k + 1] : bmax;
    }
    const bool keep_fill = !(bmin <= 0.0 && 0.0 <= bmax);  // the fill value 0 lies outside the block's range
    const double factor = (double)n / (double)rows;
    const double sigma = factor > 1.0 ? (factor - 1.0) * 0.5 : 0.0;
    int r = sigma > 0.0 ? (int)(4.0 * sigma + 0.5) : 0;
    r = r > EFP_MAXRAD ? EFP_MAXRAD : r;
    if (r > 0) {
        for (int k = tid; k <= 2 * r; k += nthreads) {
            const double xx = (double)(k - r);
            w[k] = exp(-0.5 * xx * xx / (sigma * sigma));
        }
        __syncthreads();
        double s = 0.0;
        for (int k = 0; k <= 2 * r; ++k) s += w[k];          // every thread: same order, same sum
        __syncthreads();
        for (int k = tid; k <= 2 * r; k += nthreads) w[k] = w[k] / s;
    }
    __syncthreads();
    for (int e = tid; e < rows * dim; e += nthreads) {
        const int o = e / dim, d = e - o * dim;
        double v = 0.0;
        if (n > 0) {
            const double pos = ((double)o + 0.5) * factor - 0.5;
            const double fl = floor(pos);
            const double f = pos - fl;
            const int t0 = (int)fl;
            const double a = efp_filtered(x, n, dim, d, t0, w, r);
            const double b = efp_filtered(x, n, dim, d, t0 + 1, w, r);
            v = (1.0 - f) * a + f * b;
            if (!(keep_fill && v == 0.0)) { v = v < bmin ? bmin : v; v = v > bmax ? bmax : v; }     // skimage: clip to the input's range
            if (!(v == v) || v == __builtin_inf() || v == -__builtin_inf()) v = 0.0;   // ret[isinf] = ret[isnan] = 0
        }
        out[e] = v;
    }
    __syncthreads();
}

// grid: one workgroup (256 threads) per block; blockIdx.x = global block index, track found by bisection
__global__ __launch_bounds__(256) void ef_blocks_kernel(const float *__restrict__ chroma, const int64_t *__restrict__ coff,
                                                        const float *__restrict__ mfcc, const int64_t *__restrict__ moff,
                                                        const int64_t *__restrict__ onsets, const int64_t *__restrict__ ooff,
                                                        const int64_t *__restrict__ boff, int n_tracks, EfPrepParams P,
                                                        float *__restrict__ out_mfccs, float *__restrict__ out_ssms,
                                                        float *__restrict__ out_chromas)
{
    __shared__ double xs[EFP_MAXROWS * EFP_MAXDIM];
    __shared__ double w[2 * EFP_MAXRAD + 1];
    __shared__ double colmean[EFP_MAXDIM];
    __shared__ double rownorm[EFP_MAXROWS];
    __shared__ double rowsq[EFP_MAXROWS];
    __shared__ double red[8];
    const int tid = threadIdx.x;
    const int64_t gb = blockIdx.x;
    int lo = 0, hi = n_tracks - 1;                     // track with boff[t] <= gb < boff[t + 1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (boff[mid] <= gb) lo = mid; else hi = mid - 1;
    }
    const int track = lo;
    const int b = (int)(gb - boff[track]);
    const int64_t *on = onsets + ooff[track];
    const int64_t Tm = moff[track + 1] - moff[track], Tc = coff[track + 1] - coff[track];
    auto clampi = [](int64_t v, int64_t hi_) { return v < 0 ? (int64_t)0 : (v > hi_ ? hi_ : v); };

    // ---- MFCC block: resize, zero-mean columns, unit rows
    const int R = P.mfcc_rows, C = P.ncoef;
    {
        const int64_t i1 = clampi(on[b], Tm), i2 = clampi(on[b + P.blocksize - 1], Tm);     // numpy slicing clips
        efp_resize(mfcc + moff[track] * C, i1, i2 < i1 ? i1 : i2, C, R, xs, w, red, tid, 256);
    }
    if (tid < C) {
        double s = 0.0;
        for (int o = 0; o < R; ++o) s += xs[o * C + tid];
        colmean[tid] = s / (double)R;
    }
    __syncthreads();
    for (int e = tid; e < R * C; e += 256) xs[e] -= colmean[e % C];
    __syncthreads();
    if (tid < R) {
        double s = 0.0;
        for (int d = 0; d < C; ++d) s += xs[tid * C + d] * xs[tid * C + d];
        double nr = sqrt(s);
        if (nr == 0.0) nr = 1.0;
        rownorm[tid] = nr;
    }
    __syncthreads();
    for (int e = tid; e < R * C; e += 256) {
        xs[e] = xs[e] / rownorm[e / C];
        out_mfccs[(size_t)gb * (R * C) + e] = (float)xs[e];
    }
    __syncthreads();
    // ---- SSM of the normalised rows: sqrt(max(0, |xi|^2 + |xj|^2 - 2 xi.xj)), cells with c < r
    if (tid < R) {
        double s = 0.0;
        for (int d = 0; d < C; ++d) s += xs[tid * C + d] * xs[tid * C + d];
        rowsq[tid] = s;
    }
    __syncthreads();
    {
        const int ncell = R * (R - 1) / 2;
        for (int e = tid; e < ncell; e += 256) {
            // e -> (r, c), c < r, row-major over r: e = r (r - 1) / 2 + c
            int r = (int)((1.0 + sqrt(1.0 + 8.0 * (double)e)) * 0.5);
            while (r * (r - 1) / 2 > e) --r;
            while ((r + 1) * r / 2 <= e) ++r;
            const int c = e - r * (r - 1) / 2;
            double dotv = 0.0;
            for (int d = 0; d < C; ++d) dotv += xs[r * C + d] * xs[c * C + d];
            double d2 = rowsq[r] + rowsq[c] - 2.0 * dotv;
            d2 = d2 < 0.0 ? 0.0 : d2;
            out_ssms[(size_t)gb * ncell + e] = (float)sqrt(d2);
        }
    }
    __syncthreads();
    // ---- chroma block
    {
        const int Rc = P.chroma_rows;
        const int64_t i1 = clampi(on[b], Tc), i2 = clampi(on[b + P.blocksize], Tc);
        efp_resize(chroma + coff[track] * 12, i1, i2 < i1 ? i1 : i2, 12, Rc, xs, w, red, tid, 256);
        for (int e = tid; e < Rc * 12; e += 256) out_chromas[(size_t)gb * (Rc * 12) + e] = (float)xs[e];
    }
}

// np.median over the frames of every chroma bin of every track (f32 in, f64 out): one wave per
// (track, bin); exact binary search on the order-preserving bit pattern, counted with ballots.
__global__ __launch_bounds__(64) void ef_chroma_median_kernel(const float *__restrict__ chroma, const int64_t *__restrict__ coff,
                                                              double *__restrict__ med)
{
    const int track = blockIdx.x, bin = blockIdx.y, lane = threadIdx.x;
    const int64_t T = coff[track + 1] - coff[track];
    const float *x = chroma + coff[track] * 12 + bin;
    if (T <= 0) { if (lane == 0) med[(size_t)track * 12 + bin] = __builtin_nan(""); return; }
    auto key = [](float v) {
        const unsigned u = __float_as_uint(v);
        return (u >> 31) ? ~u : (u | 0x80000000u);
    };
    auto select = [&](int64_t k) {
        unsigned lo = 0u, hi = 0xffffffffu;
        while (lo < hi) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            int64_t tot = 0;
            for (int64_t i0 = 0; i0 < T; i0 += 64) {
                const int64_t i = i0 + lane;
                const bool le = i < T && key(x[i * 12]) <= mid;
                tot += __popcll(__ballot(le));
            }
            if (tot >= k + 1) hi = mid; else lo = mid + 1;
        }
        const unsigned u = (lo >> 31) ? (lo & 0x7fffffffu) : ~lo;
        return (double)__uint_as_float(u);
    };
    const double a = select((T - 1) / 2);
    const double b = (T & 1) ? a : select(T / 2);
    // np.median of float32 data is float32: the mean of the two middle values is taken in f32
    if (lane == 0) med[(size_t)track * 12 + bin] = (T & 1) ? a : (double)(((float)a + (float)b) / 2.0f);
}

// unit-norm chroma rows in place (X / |X| with zero norms -> 1, cross_recurrence.py:66-71) and the
// squared row norms of the Euclidean features (np.sum(X**2, 1), :46): one wave per block row.
__global__ __launch_bounds__(256) void ef_rownorm_kernel(float *__restrict__ feat, int64_t nrows, int dim, int normalise,
                                                         float *__restrict__ sq)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nrows) return;
    float *x = feat + row * dim;
    double acc = 0.0;
    for (int e = lane; e < dim; e += 64) acc += (double)x[e] * (double)x[e];
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (normalise) {
        float nr = (float)sqrt(acc);
        if (nr == 0.0f) nr = 1.0f;
        for (int e = lane; e < dim; e += 64) x[e] = x[e] / nr;
    } else if (lane == 0) {
        sq[row] = (float)acc;
    }
}

}  // namespace acx
