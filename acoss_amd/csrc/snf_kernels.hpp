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

}  // namespace acx
