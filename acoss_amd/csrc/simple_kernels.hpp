// SiMPle device kernel for gfx950 (reference: acoss/algorithms/simple_silva.py:45-54, 68-126).
//
// Per ORDERED pair (i, j) (the reference runs permutations, coverid.py:135):
//   OTI      shift = last argmax_s <sum_t A, roll(sum_t B, s)>       simple_silva.py:45-54
//   profile  MP[a] = min_b  sum_{c<12, k<L} (A[c,a+k] - B'[c,b+k])^2  simple_silva.py:68-116
//            evaluated like the reference as |a|^2 + |b|^2 - 2 <a, b>, all in f64
//   score    D[i, j] = -median(MP)                                    simple_silva.py:118,125
//
// <a, b> for subsequences starting at (a, b) is the length-L window sum, along the diagonal
// b - a, of the frame Gram G[t][u] = sum_c A[c,t] B'[c,u] -- the reference's STOMP update
// (simple_silva.py:107-110) walks exactly these diagonals.  One workgroup per pair: both
// tracks live in LDS (time-major, 12 f64 per frame), thread d walks diagonal d keeping the
// last L Gram values in registers (window sum re-added from scratch every step: no drift),
// row minima are merged with 64-bit LDS atomics on order-preserving keys, and the median is
// taken by rank counting.  The whole pair is ~40 KB: the kernel is LDS/VALU resident, HBM
// traffic is the two feature reads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acx {

constexpr int SIMPLE_MAXN = 512;     // pooled frames per track supported on the device
constexpr int SIMPLE_MAXL = 16;      // subsequence length

__device__ __forceinline__ unsigned long long f64_key(double v)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);     // monotone: smaller double -> smaller key
}
__device__ __forceinline__ double key_f64(unsigned long long k)
{
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// pool: (sum n_t, 12) f64 time-major; prof: (n_tracks, 12) f64 = sum over time of every bin
template <int L>
__global__ __launch_bounds__(256) void simple_kernel(const double *__restrict__ pool,
                                                     const int64_t *__restrict__ toff,
                                                     const double *__restrict__ prof,
                                                     const int32_t *__restrict__ pairs,
                                                     double *__restrict__ out, int do_oti)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int ti = pairs[2 * blockIdx.x], tj = pairs[2 * blockIdx.x + 1];
    const int na = (int)(toff[ti + 1] - toff[ti]), nb = (int)(toff[tj + 1] - toff[tj]);
    const int ma = na - L + 1, mb = nb - L + 1;      // profile lengths
    int &s_shift = *reinterpret_cast<int *>(smem_raw);          // first 32 bytes: scalars
    double *s_med = reinterpret_cast<double *>(smem_raw + 8);
    double *A = reinterpret_cast<double *>(smem_raw + 32);     // na x 12
    double *B = A + (size_t)na * 12;                            // nb x 12 (rolled)
    double *a2 = B + (size_t)nb * 12;                           // ma
    double *b2 = a2 + ma;                                       // mb
    unsigned long long *mp = reinterpret_cast<unsigned long long *>(b2 + mb);   // ma keys

    // ---- OTI (simple_silva.py:45-54): v[s] = <pa, roll(pb, s)>; np.argsort(v)[-1]
    if (tid == 0) {
        const double *pa = prof + (size_t)ti * 12, *pb = prof + (size_t)tj * 12;
        int best = 0;
        double bestv = 0.0;
        for (int s = 0; s < 12; ++s) {
            double acc = 0.0;
            for (int c = 0; c < 12; ++c) acc += pa[c] * pb[(c - s + 12) % 12];
            if (s == 0 || acc >= bestv) { bestv = acc; best = s; }     // ties: highest index (stable sort order)
        }
        s_shift = do_oti ? best : 0;
    }
    __syncthreads();
    const int shift = s_shift;
    const double *ga = pool + toff[ti] * 12, *gb = pool + toff[tj] * 12;
    for (int idx = tid; idx < na * 12; idx += 256) A[idx] = ga[idx];
    for (int idx = tid; idx < nb * 12; idx += 256) {
        const int t = idx / 12, c = idx - 12 * t;
        int cs = c + shift; if (cs >= 12) cs -= 12;
        B[t * 12 + cs] = gb[idx];                  // np.roll(seq_b, shift, axis=0): bin c -> c + shift
    }
    __syncthreads();
    // ---- windowed squared norms
    for (int i = tid; i < ma; i += 256) {
        double acc = 0.0;
        for (int k = 0; k < L; ++k)
            for (int c = 0; c < 12; ++c) acc += A[(i + k) * 12 + c] * A[(i + k) * 12 + c];
        a2[i] = acc;
        mp[i] = 0xffffffffffffffffull;
    }
    for (int j = tid; j < mb; j += 256) {
        double acc = 0.0;
        for (int k = 0; k < L; ++k)
            for (int c = 0; c < 12; ++c) acc += B[(j + k) * 12 + c] * B[(j + k) * 12 + c];
        b2[j] = acc;
    }
    __syncthreads();
    // ---- diagonals d = b - a in [-(ma-1), mb-1]
    const int ndiag = ma + mb - 1;
    for (int dd = tid; dd < ndiag; dd += 256) {
        const int d = dd - (ma - 1);
        int a = d < 0 ? -d : 0, b = d < 0 ? 0 : d;
        // Gram values of the first window
        double g[L];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 12; ++c) acc += A[(a + k) * 12 + c] * B[(b + k) * 12 + c];
            g[k] = acc;
        }
        while (true) {
            double dot = 0.0;
#pragma unroll
            for (int k = 0; k < L; ++k) dot += g[k];
            const double dist = b2[b] + a2[a] - 2.0 * dot;
            atomicMin(&mp[a], f64_key(dist));
            ++a; ++b;
            if (a >= ma || b >= mb) break;
            // slide: drop G[a-1][b-1], append G[a+L-1][b+L-1]
#pragma unroll
            for (int k = 0; k < L - 1; ++k) g[k] = g[k + 1];
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 12; ++c) acc += A[(a + L - 1) * 12 + c] * B[(b + L - 1) * 12 + c];
            g[L - 1] = acc;
        }
    }
    __syncthreads();
    // ---- median by rank counting (np.median: mean of the two middle values for even counts)
    const int r_lo = (ma - 1) / 2, r_hi = ma / 2;
    for (int i = tid; i < ma; i += 256) {
        const unsigned long long me = mp[i];
        int rank = 0;
        for (int k = 0; k < ma; ++k) {
            const unsigned long long o = mp[k];
            rank += (o < me || (o == me && k < i)) ? 1 : 0;
        }
        if (rank == r_lo) s_med[0] = key_f64(me);
        if (rank == r_hi) s_med[1] = key_f64(me);
    }
    __syncthreads();
    if (tid == 0) {
        const double med = (r_lo == r_hi) ? s_med[0] : (s_med[0] + s_med[1]) * 0.5;
        out[blockIdx.x] = -med;
    }
}

}  // namespace acx
