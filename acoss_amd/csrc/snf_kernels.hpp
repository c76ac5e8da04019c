// Similarity network fusion, the N x N late-fusion post-step of EarlyFusion / LateFusionChen
// (acoss/algorithms/utils/similarity_fusion.py:146-186, doSimilarityFusionWs): per sweep and
// matrix i,  P_i <- S_i (mean_{k != i} P_k) S_i^T (+ reg on the diagonal),  S_i the row-normalised
// K-nearest-neighbour kernel of W_i (K non-zeros per row).  The reference multiplies scipy sparse
// matrices on the host; at N = 15 000 that post-step takes longer than the whole pair grid on 8
// GPUs.  Here: f64 like the reference, S_i as (index, weight) lists, two gather kernels per update.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acx {

// P = W / rowsum(W) (rows that sum to 0 stay as they are); one workgroup per row
__global__ __launch_bounds__(256) void snf_rownorm_kernel(const double *__restrict__ W, double *__restrict__ P, int n)
{
    __shared__ double part[256];
    const int i = blockIdx.x;
    const double *w = W + (size_t)i * n;
    double s = 0.0;
    for (int c = threadIdx.x; c < n; c += 256) s += w[c];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    double rs = part[0];
    if (rs == 0.0) rs = 1.0;
    double *p = P + (size_t)i * n;
    for (int c = threadIdx.x; c < n; c += 256) p[c] = w[c] / rs;
}

// acc = (sum of the given matrices) * scale, elementwise; srcs: up to 8 pointers
struct SnfSrc { const double *p[8]; int count; };
__global__ __launch_bounds__(256) void snf_mean_kernel(SnfSrc src, double scale, double *__restrict__ acc, int64_t total)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    double s = 0.0;
    for (int k = 0; k < src.count; ++k) s += src.p[k][e];
    acc[e] = s * scale;
}

// UT = A S^T:  UT[j][i] = sum_k V[i][k] A[j][J[i][k]].  One workgroup per row j; the row of A sits in
// LDS (n <= 20 000 doubles) and every thread gathers its K neighbours from it.  LDSROW = false:
// the gathers go to global memory (any n).
template <bool LDSROW>
__global__ __launch_bounds__(256) void snf_ast_kernel(const double *__restrict__ A, const int32_t *__restrict__ J,
                                                      const double *__restrict__ V, double *__restrict__ UT, int n, int K)
{
    extern __shared__ double row[];
    const int j = blockIdx.x;
    const double *a = A + (size_t)j * n;
    if (LDSROW) {
        for (int c = threadIdx.x; c < n; c += 256) row[c] = a[c];
        __syncthreads();
    }
    const double *src = LDSROW ? row : a;
    double *ut = UT + (size_t)j * n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int32_t *ji = J + (size_t)i * K;
        const double *vi = V + (size_t)i * K;
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += vi[k] * src[ji[k]];
        ut[i] = s;
    }
}

// P = S UT (+ reg on the diagonal):  P[i][c] = sum_k V[i][k] UT[J[i][k]][c].  One workgroup per row i,
// threads over the columns: K coalesced row reads per output row.
__global__ __launch_bounds__(256) void snf_sut_kernel(const double *__restrict__ UT, const int32_t *__restrict__ J,
                                                      const double *__restrict__ V, double *__restrict__ P, int n, int K,
                                                      double reg_diag)
{
    const int i = blockIdx.x;
    const int32_t *ji = J + (size_t)i * K;
    const double *vi = V + (size_t)i * K;
    double *p = P + (size_t)i * n;
    for (int c = threadIdx.x; c < n; c += 256) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += vi[k] * UT[(size_t)ji[k] * n + c];
        if (c == i) s += reg_diag;
        p[c] = s;
    }
}

// ------------------------------------------------------------------------------------
// Affinity matrices and neighbour lists on the device (getW, similarity_fusion.py:15-36; getS,
// :124-144): the host only hands over the N x N distance matrices.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long snf_key(double v)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);     // monotone: smaller double -> smaller key
}
__device__ __forceinline__ double snf_unkey(unsigned long long k)
{
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// S = 0.5 (D + D^T) with a zero diagonal; 32 x 32 tiles through LDS so that both reads are coalesced
__global__ __launch_bounds__(256) void snf_sym_kernel(const double *__restrict__ D, double *__restrict__ S, int n)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int i = bx + r, j = by + tx;                            // D[bx + r][by + tx]: the transposed tile
        tile[r][tx] = (i < n && j < n) ? D[(size_t)i * n + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = by + r, j = bx + tx;
        if (i < n && j < n) {
            const double v = 0.5 * (D[(size_t)i * n + j] + tile[tx][r]);
            S[(size_t)i * n + j] = (i == j) ? 0.0 : v;
        }
    }
}

// k-th smallest (0-based) key of a row of n doubles: binary search on the key, ballots count
__device__ __forceinline__ unsigned long long snf_select(const double *__restrict__ row, int n, int k, int lane)
{
    unsigned long long lo = 0ull, hi = 0xffffffffffffffffull;
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int tot = 0;
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int c = c0 + lane;
            const bool le = c < n && snf_key(row[c]) <= mid;
            tot += __popcll(__ballot(le));
        }
        if (tot >= k + 1) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// md[i] = mean of the K + 1 smallest of row i of S (the zero diagonal included) * (K + 1) / K
// (similarity_fusion.py:27-30).  One wave per row; ties at the cut contribute their common value.
__global__ __launch_bounds__(256) void snf_localscale_kernel(const double *__restrict__ S, double *__restrict__ md, int n, int K)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const double *row = S + (size_t)i * n;
    const int take = K + 1 < n ? K + 1 : n;
    const unsigned long long tk = snf_select(row, n, take - 1, lane);
    const double t = snf_unkey(tk);
    double s = 0.0;
    int below = 0;
    for (int c = lane; c < n; c += 64) {
        const double v = row[c];
        if (snf_key(v) < tk) { s += v; ++below; }
    }
    for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off, 64); below += __shfl_xor(below, off, 64); }
    if (lane == 0) md[i] = (s + (double)(take - below) * t) / (double)take * (double)(K + 1) / (double)K;
}

// W = exp(-S^2 / (2 (mu eps)^2)), eps = (md_i + md_j + S_ij) / 3, zero denominators -> 1 (:31-36); in place
__global__ __launch_bounds__(256) void snf_affinity_kernel(double *__restrict__ S, const double *__restrict__ md, int n, double mu)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * n) return;
    const int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
    const double d = S[e];
    const double eps = (md[i] + md[j] + d) / 3.0;
    double den = 2.0 * (mu * eps) * (mu * eps);
    if (den == 0.0) den = 1.0;
    S[e] = exp(-(d * d) / den);
}

// The K largest of every row of W (ties at the cut in column order -- a stable descending sort, like
// the host's _knn_lists), ordered by (value descending, column), weights divided by their sum (getS).
// One wave per row, K <= 64.
__global__ __launch_bounds__(256) void snf_knn_kernel(const double *__restrict__ W, int32_t *__restrict__ J, double *__restrict__ V,
                                                      int n, int K)
{
    __shared__ double cv[4][64];
    __shared__ int cj[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wv;
    if (i >= n) return;
    const double *row = W + (size_t)i * n;
    // K-th largest = (n - K)-th smallest (0-based)
    const unsigned long long tk = snf_select(row, n, n - K, lane);
    int ngt = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + lane;
        ngt += __popcll(__ballot(c < n && snf_key(row[c]) > tk));
    }
    int neq_left = K - ngt;          // ties taken in column order
    int filled = 0;
    for (int c0 = 0; c0 < n && filled < K; c0 += 64) {
        const int c = c0 + lane;
        const unsigned long long k = c < n ? snf_key(row[c]) : 0ull;
        const bool gt = c < n && k > tk;
        const unsigned long long meq = __ballot(c < n && k == tk);
        const int eq_rank = __popcll(meq & ((1ull << lane) - 1ull));
        const bool eq = c < n && k == tk && eq_rank < neq_left;
        const unsigned long long m = __ballot(gt || eq);
        if (gt || eq) {
            const int pos = filled + __popcll(m & ((1ull << lane) - 1ull));
            cv[wv][pos] = row[c];
            cj[wv][pos] = c;
        }
        filled += __popcll(m);
        const int eq_taken = __popcll(meq) < neq_left ? __popcll(meq) : neq_left;
        neq_left -= eq_taken;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double v = lane < K ? cv[wv][lane] : 0.0;
    const int jj = lane < K ? cj[wv][lane] : 0;
    int rank = 0;
    for (int t = 0; t < K; ++t) {
        const double o = cv[wv][t];
        const int oj = cj[wv][t];
        rank += (o > v || (o == v && oj < jj)) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    double sum = 0.0;
    for (int t = 0; t < K; ++t) sum += cv[wv][t];     // (any order of the K terms: the same values)
    if (sum == 0.0) sum = 1.0;
    if (lane < K) {
        J[(size_t)i * K + rank] = jj;
        V[(size_t)i * K + rank] = v / sum;
    }
}

}  // namespace acx
