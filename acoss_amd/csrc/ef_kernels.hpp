// EarlyFusion per-pair device kernels for gfx950 (reference:
// acoss/algorithms/earlyfusion_traile.py:157-198 and the L2 utilities it calls).
//
//   E1 ef_gemm_kernel   the three cross-similarity matrices of a pair as f32 MFMA GEMMs
//                       (v_mfma_f32_16x16x4_f32, K = 650 / 1225 / 480) with the epilogue of
//                       get_csm (cross_recurrence.py:30-48: sqrt(max(0, |x|^2+|y|^2-2xy))) or
//                       get_csm_cosine after the blocked OTI roll (:53-73, :105-134: rows are
//                       pre-normalised at upload, the roll is a permutation of the k index);
//                       writes C and C^T; the chroma matrix takes this kernel, the two Euclidean ones
//   E1b ef_gemm_bf16x3_kernel  the same products from three-term bf16 splits on the bf16 matrix pipe
//   E2 ef_rowstat_kernel one wave per row: the k-th smallest value of the row (threshold of
//                       csm_to_binary, :136-161) and the mean of the K smallest
//                       (getWCSM, similarity_fusion.py:46-50); on C^T rows = column stats
//   E3 ef_fuse_kernel   fused = exp(-(W_mfcc + W_ssm + W_chroma)),
//                       W = exp(-C^2 / (2 (0.5 (r_i + c_j + C)/3)^2))   (similarity_fusion.py:51-54,
//                       earlyfusion_traile.py:178-182)
//   E4 sw_kernel        constrained Smith-Waterman (alignment_tools.py:26-46) as an exact
//                       integer DP in tenths (+10 / -10 / -7), binarising on the fly
//                       (B_ij = C_ij <= t_i), one wave per matrix, row sweep with the two
//                       previous rows in registers
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "serra09_kernels.hpp"

namespace acx {

constexpr int EF_COLSTAT_MAXK = 16;   // neighbourhood sizes K the column statistics take from C itself (ef_colstat_kernel); beyond: C^T + row kernels
constexpr int EF_MAXNB = 1024;    // blocks per track of the register-resident kernels (rows of <= 512 take the narrow ones);
                                  // longer tracks: ef_rowstat_long_kernel / sw_long_kernel (any length)

struct EfPair {
    int32_t q, r;          // track indices
    int32_t M, N;          // blocks of q / r
    int32_t oti;           // get_oti(chroma_med_q, chroma_med_r)
    int32_t pitchC;        // row pitch of C (multiple of 64, >= N)
    int32_t pitchT;        // row pitch of C^T (multiple of 64, >= M)
    int32_t kbin;          // neighbours per row of csm_to_binary (host: int(round(kappa * N)), half-to-even)
    int32_t ctN;           // N when the transposed matrices are kept (neighbourhoods of more than EF_COLSTAT_MAXK columns), else 0
    int32_t pad;
    int64_t offB;          // word offset of the pair's four binarised matrices (mfccs, ssms, chromas, fused): M rows of pitchC / 32 words
    int64_t offC;          // float offset of the pair's matrices: [C x3][C^T x3 (ctN rows each)][F]
    int64_t offS;          // float offset of the pair's vectors:
                           //   per feature s<3: [t rows][r rows][c cols][jcut rows]; then [t rows][jcut rows] of F
};

__device__ __forceinline__ int64_t ef_c_off(const EfPair &P, int s) { return P.offC + (int64_t)s * P.M * P.pitchC; }
__device__ __forceinline__ int64_t ef_ct_off(const EfPair &P, int s)
{
    return P.offC + (int64_t)3 * P.M * P.pitchC + (int64_t)s * P.ctN * P.pitchT;
}
__device__ __forceinline__ int64_t ef_f_off(const EfPair &P)
{
    return P.offC + (int64_t)3 * P.M * P.pitchC + (int64_t)3 * P.ctN * P.pitchT;
}
// vectors: feature s: t at +0, r at +pitchT, c at +2 pitchT (size pitchC), jcut (int) behind c; stride
// per feature.  The fused matrix (4th slot) keeps t at +0 and jcut at +pitchT.
// Binarisation rule (csm_to_binary keeps exactly k cells per row): B_ij = C_ij < t_i, or C_ij == t_i
// and j <= jcut_i -- ties at the k-th value are taken in column order until k cells are set.
__host__ __device__ __forceinline__ int64_t ef_s_stride(const EfPair &P) { return 3 * (int64_t)P.pitchT + P.pitchC; }
// strip-boundary records of the long Smith-Waterman: behind the four vector blocks, 4 pitchT ints per matrix
__host__ __device__ __forceinline__ int64_t ef_rec_off(const EfPair &P, int src) { return 4 * ef_s_stride(P) + (int64_t)src * 4 * P.pitchT; }
__host__ __device__ __forceinline__ int64_t ef_s_total(const EfPair &P) { return 4 * ef_s_stride(P) + 16 * (int64_t)P.pitchT; }
__host__ __device__ __forceinline__ int64_t ef_jcut_off(const EfPair &P, int src)
{
    return src < 3 ? 2 * (int64_t)P.pitchT + P.pitchC : (int64_t)P.pitchT;
}

// sqrt of a distance^2 (>= 0) for get_csm's epilogues: v_sqrt_f32 (1 ulp) and one Newton step on the fma residual --
// the correctly rounded value except in rare halfway cases, never more than 1 ulp off (tests/test_gpu_earlyfusion.py::
// test_epilogue_sqrt), in 9 instructions.  The compiler's IEEE expansion of sqrtf is ~20 (input scaling for denormals,
// two integer-stepped candidates, class tests): 64 of them per lane made the rectangle GEMM's epilogue VALU-bound
// (1 900 VALU instructions per wave = 15 k of a tile's 87 k cycles).  Zero, denormal results and inf pass through.
__device__ __forceinline__ float ef_sqrt_nonneg(float x)
{
    const float r = __builtin_amdgcn_sqrtf(x);
    const float e = __builtin_fmaf(-r, r, x);
    const float c = __builtin_fmaf(e, 0.5f * __builtin_amdgcn_rcpf(r), r);
    return (r > 1e-18f && r < 1e18f) ? c : r;
}
static __global__ void ef_sqrt_probe_kernel(const float *in, float *out, int64_t n)       // (acx_debug_ef_sqrt)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ef_sqrt_nonneg(in[i]);
}

// ---- OTI of the pair (cross_recurrence.py:75-103): argmax_s sum(roll(C1, s) * C2), f64, first max
__global__ void ef_oti_kernel(EfPair *pd, int B, const double *__restrict__ med)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    const double *c1 = med + (size_t)12 * pd[p].q, *c2 = med + (size_t)12 * pd[p].r;
    int best = 0;
    double bestv = 0.0;
    for (int s = 0; s < 12; ++s) {
        double acc = 0.0;
        for (int c = 0; c < 12; ++c) acc += c1[(c - s + 12) % 12] * c2[c];
        if (s == 0 || acc > bestv) { bestv = acc; best = s; }
    }
    pd[p].oti = best;
}

// epilogue + stores of a wave's 64 x 64 block of accumulators (acc[a][b][reg] = dot of row ib + 16 a + 4 lk + reg
// and column jb + 16 b + lr): get_csm / get_csm_cosine, C row-major (16 lanes in a row write 64 contiguous
// bytes) and C^T (a lane's four accumulator rows are four consecutive columns of C^T: one 16-byte store)
template <int NA>
__device__ __forceinline__ void ef_gemm_epilogue(const f32x4 (&acc)[NA][4], const EfPair &P, int s, int ib0, int jb0, int na, int nb,
                                                 int lr, int lk, const float *__restrict__ nrm, const int64_t *__restrict__ boff,
                                                 float *__restrict__ scratch)
{
    float *C = scratch + ef_c_off(P, s);
    float *CT = scratch + ef_ct_off(P, s);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (!(a < na && b < nb)) continue;
            const int ib = ib0 + 16 * a + 4 * lk;
            const int j = jb0 + 16 * b + lr;
            float v[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = ib + reg;
                const float dot = acc[a][b][reg];
                if (s == 2) {
                    v[reg] = 1.0f - dot;
                } else {
                    const float nx = (i < P.M) ? nrm[boff[P.q] + i] : 0.0f, ny = (j < P.N) ? nrm[boff[P.r] + j] : 0.0f;
                    float t = (nx + ny) - 2.0f * dot;
                    if (t < 0.0f) t = 0.0f;
                    v[reg] = ef_sqrt_nonneg(t);
                }
                if (i < P.M && j < P.N) C[(size_t)i * P.pitchC + j] = v[reg];
            }
            if (j < P.N && P.ctN) {
                float *ct = CT + (size_t)j * P.pitchT + ib;
                if (ib + 3 < P.M) *reinterpret_cast<float4 *>(ct) = make_float4(v[0], v[1], v[2], v[3]);
                else
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        if (ib + reg < P.M) ct[reg] = v[reg];
            }
        }
}

// ------------------------------------------------------------------------------------
// E1: C[i][j] = epilogue( sum_k A[i][perm(k)] * B[j][k] ), 128 x 128 tile per workgroup,
// 4 waves as 2 x 2, each wave 64 x 64 = 4 x 4 MFMA tiles (16 accumulator tiles: every operand
// read from LDS feeds four MFMAs -- an f32 MFMA blocks VALU / LDS issue on its SIMD, so the
// instructions AROUND the MFMAs are what the GEMM loses time to).  K is walked in blocks of 24 (two
// 12-bin groups of the chroma roll): the next block's 16-byte global loads are in flight in registers
// while the current one is multiplied out of LDS (k-major, conflict-free operand reads), two
// barriers per 24 k.  16 x 16 sub-tiles that lie entirely outside the matrix are skipped (wave-uniform),
// so padding costs at most 15 rows / columns.  The blocked-OTI roll of the first song's chroma
// is applied as a permutation of the LDS k-row on the way in.
// feat: 0 mfcc (euclid), 1 ssm (euclid), 2 chroma (cosine, A rolled by oti).
// ------------------------------------------------------------------------------------
constexpr int EF_BK = 24;
constexpr int EF_TILE = 128;
constexpr int EF_LP = 144;     // LDS pitch (k-major, 128 rows + pad; 144 % 32 == 16)

__global__ __launch_bounds__(256) void ef_gemm_kernel(const float *__restrict__ feat0, const float *__restrict__ feat1,
                                                      const float *__restrict__ feat2,
                                                      const float *__restrict__ nrm0, const float *__restrict__ nrm1,
                                                      const int64_t *__restrict__ boff,
                                                      const EfPair *__restrict__ pd, float *__restrict__ scratch,
                                                      int K0, int K1, int K2, int tiles_x, int s0)
{
    __shared__ float As[EF_BK * EF_LP];
    __shared__ float Bs[EF_BK * EF_LP];
    const EfPair P = pd[blockIdx.y];
    const int s = s0 + blockIdx.z;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int i0 = ty * EF_TILE, j0 = tx * EF_TILE;
    if (i0 >= P.M || j0 >= P.N) return;
    const int K = s == 0 ? K0 : (s == 1 ? K1 : K2);
    const float *F = s == 0 ? feat0 : (s == 1 ? feat1 : feat2);
    const int rot = (s == 2) ? P.oti : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    // 16-row / 16-column sub-tiles of this wave that hold at least one cell of the matrix
    int na = (P.M - (i0 + 64 * wr) + 15) / 16, nb = (P.N - (j0 + 64 * wc) + 15) / 16;
    na = na < 0 ? 0 : (na > 4 ? 4 : na);
    nb = nb < 0 ? 0 : (nb > 4 ? 4 : nb);

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: thread -> (row = tid / 2, 12 consecutive k = 12 * (tid % 2) ...) as three float4 per operand
    const int srow = tid >> 1, sk = (tid & 1) * 12;
    const bool rowa = i0 + srow < P.M, rowb = j0 + srow < P.N;
    const float *ap = F + (boff[P.q] + (rowa ? i0 + srow : 0)) * K + sk;      // (rows past the matrix read row 0; never stored)
    const float *bp = F + (boff[P.r] + (rowb ? j0 + srow : 0)) * K + sk;
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    float ra[12], rb[12];
    auto gload_full = [&]() {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const f32x4 va = *reinterpret_cast<const f32x4u *>(ap + 4 * q);
            const f32x4 vb = *reinterpret_cast<const f32x4u *>(bp + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) { ra[4 * q + e] = va[e]; rb[4 * q + e] = vb[e]; }
        }
        ap += EF_BK;
        bp += EF_BK;
    };
    auto gload_tail = [&](int k0) {                   // the block that crosses K: element-wise, zeros beyond K
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const bool ok = k0 + sk + e < K;
            ra[e] = ok ? ap[e] : 0.f;
            rb[e] = ok ? bp[e] : 0.f;
        }
    };
    auto gload = [&](int k0) {
        if (k0 + EF_BK <= K) gload_full();            // workgroup-uniform
        else gload_tail(k0);
    };
    // k-rows of the second 12-group keep their 128 rows XORed with 16: the two threads that stage one
    // row then land in different bank groups (12 * 144 = 0 mod 32), the operand reads (one group per
    // MFMA step) stay conflict free
    const int swrow = srow ^ (16 * (tid & 1));
    float *as0 = As + sk * EF_LP + swrow, *bs0 = Bs + sk * EF_LP + swrow;
    auto lstore = [&]() {
        if (rot == 0) {                                // workgroup-uniform: all offsets are immediates
#pragma unroll
            for (int e = 0; e < 12; ++e) { as0[e * EF_LP] = ra[e]; bs0[e * EF_LP] = rb[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                // A[k] multiplies B[k'] with k' = k - c + (c + rot) mod 12, c = k mod 12 = e here
                int ea = e + rot; ea = ea >= 12 ? ea - 12 : ea;
                as0[ea * EF_LP] = ra[e];
                bs0[e * EF_LP] = rb[e];
            }
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += EF_BK) {
        lstore();
        __syncthreads();
        if (k0 + EF_BK < K) gload(k0 + EF_BK);           // in flight during the MFMAs below
#pragma unroll
        for (int kb = 0; kb < EF_BK / 4; ++kb) {
            const int sw = 16 * (kb / 3);                 // the XOR of this k-group
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = As[(4 * kb + lk) * EF_LP + ((64 * wr + 16 * a + lr) ^ sw)];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = Bs[(4 * kb + lk) * EF_LP + ((64 * wc + 16 * b + lr) ^ sw)];
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < na) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (b < nb) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    ef_gemm_epilogue<4>(acc, P, s, i0 + 64 * wr, j0 + 64 * wc, na, nb, lr, lk, s == 0 ? nrm0 : nrm1, boff, scratch);
}

// ------------------------------------------------------------------------------------
// E1b: the two Euclidean cross-similarity matrices (mfcc, ssm: 80 % of the chain's flops) on the bf16 matrix
// pipe, at f32 accuracy.  Every f32 feature value is split ONCE per pool into three bf16 terms
// x = x1 + x2 + x3 (round-to-nearest each, remainders exact: 3 x 8 significant bits), and
//   x . y  ~=  x1 y1 + (x1 y2 + x2 y1) + (x2 y2 + x1 y3 + x3 y1)
// (the dropped terms are below 2^-24 |x| |y|, the size of one f32 rounding) with f32 accumulation in the
// MFMA, smallest terms first.  Six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block cost 6 x 16 cycles where
// eight v_mfma_f32_16x16x4_f32 cost 8 x 32 -- and unlike the f32 MFMA the bf16 MFMA leaves VALU / LDS issue
// of the other waves alone (scripts/ubench/mfma_valu_overlap.hip).  Same tiling as ef_gemm_kernel
// (128 x 128 per workgroup) but 8 waves of 32 x 64 cells, k in blocks of 32: a lane's MFMA operand is 8
// consecutive k of one row = one 16-byte LDS read (XOR-swizzled 64-byte rows: conflict free), 18 operand reads feed 48 MFMAs.
// Layout of the split pool: [block][k / 32][term 0..2][k % 32] bf16, Kp = K rounded up to 32, zeros behind K.
// ------------------------------------------------------------------------------------
constexpr int EFB_BK = 32;
constexpr int EFB_LP = 32;     // LDS row pitch in bf16 elements (64 bytes, no padding): the four 16-byte pieces of a row are
                               // stored XOR-swizzled, piece ^ ((row >> 2) & 3) -- the staging stores (4 rows x 4 pieces per 16 lanes)
                               // and the operand reads (16 rows x 1 piece per 16 lanes) are then both bank-conflict free (round 3:
                               // with a padded pitch of 80 bytes half of the kernel's LDS cycles were conflicts of the stores)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned ef_bf16_rne(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// ACX_EF_GEMM_F16X2: the power-of-two scale of every pool row (one wave per row).  inv[row] = 2^(e - 15) with
// max |x| of the row = f 2^e, f in [0.5, 1): scale * x = x / inv lies in [2^14, 2^15) for the row's largest value, so
// the second fp16 term of every value within 2^-17 of it is a normal number (smaller ones lose bits that are below
// 2^-39 of the row's largest value -- nothing an f32 dot product of the row keeps either).  The scale depends on the
// ROW alone: a track's operands, and with them a pair's score, do not depend on what else is in the pool.
__global__ __launch_bounds__(256) void ef_rowscale_kernel(const float *__restrict__ f, float *__restrict__ inv, int64_t nblocks, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nblocks) return;
    unsigned m = 0;
    for (int k = lane; k < K; k += 64) {
        const unsigned u = __float_as_uint(f[row * K + k]) & 0x7fffffffu;       // (non-negative floats order like their bits)
        m = u > m ? u : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned v = (unsigned)__shfl_xor((int)m, o, 64);
        m = v > m ? v : m;
    }
    if (lane == 0) {
        int e = (int)(m >> 23) - 126;                                           // frexp's exponent (subnormal / zero rows: clamped below)
        e = e < -95 ? -95 : (e > 110 ? 110 : e);
        inv[row] = __uint_as_float((unsigned)(e - 15 + 127) << 23);
    }
}

// (grid-stride: a launch carries at most 2^32 - 1 work-items per dimension, and a DA-TACOS-sized pool has 7.5e9
// split values per feature -- a one-thread-per-value grid wraps silently and leaves the tail of the pool unwritten)
__global__ __launch_bounds__(256) void ef_split_bf16_kernel(const float *__restrict__ f, unsigned short *__restrict__ out,
                                                            int64_t nblocks, int K, int Kp, int binmajor, const float *__restrict__ inv)
{
    const int64_t total = nblocks * Kp, stride = (int64_t)gridDim.x * 256;
    const int G = K / 12;                                             // (binmajor: frames of a chroma block)
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
        const int64_t row = idx / Kp;
        const int k = (int)(idx - row * Kp);
        // binmajor (chroma, K = 12 G): position k = G bin + frame of the split row takes value 12 frame + bin of the block
        const int ks = binmajor ? 12 * (k % G) + k / G : k;
        const float x = k < K ? f[row * K + ks] : 0.0f;
        unsigned h1, h2, h3;
        const int nt = inv ? 2 : 3;                                   // terms kept per value: a row and 32-k chunk is 128 / 192 contiguous bytes
        if (!inv) {
            h1 = ef_bf16_rne(x);
            const float r1 = x - __uint_as_float(h1 << 16);            // exact
            h2 = ef_bf16_rne(r1);
            const float r2 = r1 - __uint_as_float(h2 << 16);           // exact
            h3 = ef_bf16_rne(r2);
        } else {
            // ACX_EF_GEMM_F16X2: two fp16 terms of x / inv[row] (ef_rowscale_kernel; a power of two: its reciprocal by
            // the exponent bits), 128 contiguous bytes = one cache line per row and 32-k chunk
            const float xs = x * __uint_as_float(0x7F000000u - __float_as_uint(inv[row]));      // exact
            const _Float16 a = (_Float16)xs;
            const _Float16 b = (_Float16)(xs - (float)a);              // (the remainder is exact in f32)
            h1 = __builtin_bit_cast(unsigned short, a);
            h2 = __builtin_bit_cast(unsigned short, b);
            h3 = 0;
        }
        // [block][k / 32][term][k % 32]: the three terms of a 32-k chunk are 192 contiguous bytes, so the GEMM's three
        // 64-byte reads per row and chunk share their 128-byte lines
        unsigned short *o = out + row * nt * Kp + (int64_t)(k / EFB_BK) * (nt * EFB_BK) + (k % EFB_BK);
        o[0] = (unsigned short)h1; o[EFB_BK] = (unsigned short)h2;
        if (!inv) o[2 * EFB_BK] = (unsigned short)h3;
    }
}

constexpr int EFB_THREADS = 512;    // 8 waves as 4 x 2: 32 x 64 cells per wave (two workgroups per CU = 4 waves per SIMD)

__global__ __launch_bounds__(EFB_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void ef_gemm_bf16x3_kernel(const unsigned short *__restrict__ split0,
                                                                        const unsigned short *__restrict__ split1,
                                                                        const float *__restrict__ nrm0, const float *__restrict__ nrm1,
                                                                        const int64_t *__restrict__ boff,
                                                                        const EfPair *__restrict__ pd, float *__restrict__ scratch,
                                                                        int Kp0, int Kp1, int tiles_x)
{
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * EF_TILE * EFB_LP];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * EF_TILE * EFB_LP];
    const EfPair P = pd[blockIdx.y];
    const int s = blockIdx.z;                        // 0 mfcc, 1 ssm
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int i0 = ty * EF_TILE, j0 = tx * EF_TILE;
    if (i0 >= P.M || j0 >= P.N) return;
    const int Kp = s == 0 ? Kp0 : Kp1;
    const unsigned short *S = s == 0 ? split0 : split1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int NA = 2;                            // 16-row sub-tiles per wave
    int na = (P.M - (i0 + 32 * wr) + 15) / 16, nb = (P.N - (j0 + 64 * wc) + 15) / 16;
    na = na < 0 ? 0 : (na > NA ? NA : na);
    nb = nb < 0 ? 0 : (nb > 4 ? 4 : nb);

    f32x4 acc[NA][4];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: thread -> (row = tid / 4, 8 consecutive k = 8 * (tid % 4) ...) of every term: one 16-byte piece each
    const int srow = tid >> 2, sk = (tid & 3) * 8;
    const bool rowa = i0 + srow < P.M, rowb = j0 + srow < P.N;
    const unsigned short *ap = S + (boff[P.q] + (rowa ? i0 + srow : 0)) * 3 * Kp + sk;    // (rows past the matrix read row 0; never stored)
    const unsigned short *bp = S + (boff[P.r] + (rowb ? j0 + srow : 0)) * 3 * Kp + sk;
    u32x4 ra[3], rb[3];
    auto gload = [&]() {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            ra[t] = *reinterpret_cast<const u32x4 *>(ap + t * EFB_BK);
            rb[t] = *reinterpret_cast<const u32x4 *>(bp + t * EFB_BK);
        }
        ap += 3 * EFB_BK;
        bp += 3 * EFB_BK;
    };
    const int skl = ((tid & 3) ^ ((srow >> 2) & 3)) * 8;             // swizzled piece of the row in LDS
    unsigned short *as0 = As + srow * EFB_LP + skl, *bs0 = Bs + srow * EFB_LP + skl;
    auto lstore = [&]() {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            *reinterpret_cast<u32x4 *>(as0 + t * EF_TILE * EFB_LP) = ra[t];
            *reinterpret_cast<u32x4 *>(bs0 + t * EF_TILE * EFB_LP) = rb[t];
        }
    };
    const int lks = lk ^ ((lr >> 2) & 3);                             // (rows 16 a + lr: (row >> 2) & 3 == (lr >> 2) & 3)
    const unsigned short *aop = As + (32 * wr + lr) * EFB_LP + 8 * lks;
    const unsigned short *bop = Bs + (64 * wc + lr) * EFB_LP + 8 * lks;
    gload();
    for (int k0 = 0; k0 < Kp; k0 += EFB_BK) {
        lstore();
        __syncthreads();
        if (k0 + EFB_BK < Kp) gload();                     // in flight during the MFMAs below
        bf16x8 av[NA][3];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                av[a][t] = *reinterpret_cast<const bf16x8 *>(aop + (t * EF_TILE + 16 * a) * EFB_LP);
        // term-major, so that consecutive MFMAs go to different accumulators; a wave whose block lies inside the
        // matrix runs the unguarded sequence
        auto mma = [&](auto full_tag) {
            constexpr bool full = decltype(full_tag)::value;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (full || b < nb) {
                    bf16x8 bv[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) bv[t] = *reinterpret_cast<const bf16x8 *>(bop + (t * EF_TILE + 16 * b) * EFB_LP);
#define ACX_EFB_TERM(TA_, TB_)                                                                                         \
    _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                                     \
        if (full || a < na) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[a][TA_], bv[TB_], acc[a][b], 0, 0, 0);
                    ACX_EFB_TERM(0, 2) ACX_EFB_TERM(2, 0) ACX_EFB_TERM(1, 1) ACX_EFB_TERM(0, 1) ACX_EFB_TERM(1, 0) ACX_EFB_TERM(0, 0)
#undef ACX_EFB_TERM
                }
            }
        };
        if (na == NA && nb == 4) mma(std::true_type());
        else mma(std::false_type());
        __syncthreads();
    }
    ef_gemm_epilogue<NA>(acc, P, s, i0 + 32 * wr, j0 + 64 * wc, na, nb, lr, lk, s == 0 ? nrm0 : nrm1, boff, scratch);
}

// ------------------------------------------------------------------------------------
// E1c: the cross-similarity products for a whole RECTANGLE of pairs at once (round 3).  The per-pair kernel
// above tiles every (M x N) matrix on its own: a 400 x 400 matrix needs 16 workgroup tiles of 128 x 128 where 9.8
// would hold its cells (measured: 92 TFLOP/s f32-equivalent at 400 blocks per track against 131 - 138 at 384 / 512),
// and every pair re-reads its two tracks.  Here the blocks of up to 128 query tracks are laid end to end as the
// rows of ONE matrix and the blocks of up to 128 reference tracks as its columns (every track padded to a multiple
// of 16 rows, so that a 16 x 16 MFMA sub-tile never straddles two pairs): a dense GEMM whose tiles are full, whose
// operands are shared by all the pairs of a query / reference track, and whose epilogue drops every 16 x 16
// sub-tile into the matrix of the pair it belongs to (pairtab: -1 = that (query, reference) combination is not
// asked for -- the lower triangle of a diagonal grid tile, or an arbitrary pair list).  The arithmetic of a cell
// is untouched -- the same k chunks in the same order through the same six MFMAs --, so the matrices are
// bit-identical to the per-pair kernel's.
// ------------------------------------------------------------------------------------
struct EfSegGroup {        // 16 consecutive rows (or columns) of the laid-out matrix
    int64_t poolrow;       // first block (pool row) of the group
    int32_t valid;         // rows of the group that exist (1 .. 16; 0: padding group)
    int32_t slot;          // index of its track among the rectangle's query (reference) tracks
    int32_t local0;        // row (column) of the group's first block inside the pair's own matrix
    int32_t pad;
};
struct EfSegRect {
    int32_t g0, ng;        // row groups [g0, g0 + ng) of the batch's row-group array
    int32_t h0, nh;        // column groups
    int32_t ncols;         // reference tracks of the rectangle (row length of its pair table)
    int32_t ptab0;         // offset of its pair table (nrows x ncols ints: index into the batch's EfPair array or -1)
};
struct EfSegWg { int32_t rect, ty, tx, pad; };      // one workgroup tile that holds at least one pair (units: see the kernels)

// ------------------------------------------------------------------------------------
// E1c': the rectangle GEMM at 256 x 128 cells per workgroup (round 3, second pass).  Counters of its 128 x 128
// predecessor (E1b's loop over rectangles): bf16 matrix pipe 38 % busy, LDS 38 % busy, a k chunk takes 8 k cycles where its MFMAs need 3 k -- every wave
// of a workgroup stores, loads and multiplies in the same phase, two barriers per chunk, and the memory phases of
// one workgroup do not fill the MFMA phases of the other.  This kernel:
//   * 8 waves of 64 x 64 cells (24 operand reads feed 96 MFMAs: 2/3 of the LDS bytes per flop);
//   * ONE workgroup per CU with 256 registers per lane, LDS double-buffered (2 x 72 KB): one barrier per chunk;
//   * a chunk is ONE basic block of 24 groups of 4 MFMAs with the wave's memory instructions dealt between the
//     groups -- the LDS stores of chunk k + 1, the global loads of chunk k + 2, the operand reads of the next column
//     sub-tile -- so that the matrix pipe runs while they issue;
//   * a cell's products in the order of E1b (same k chunks, same six MFMAs): bit-identical matrices for mfcc / ssm;
//   * CH = 1: the CHROMA matrix from the same loop (round 3; f32 MFMAs before).  Chroma rows are unit vectors of
//     G frames x 12 bins, and the first song's bins are rolled by the pair's OTI (get_csm_blocked_oti,
//     cross_recurrence.py:105-134).  The split pool keeps them BIN-major (k'' = G bin + frame): the roll becomes a
//     cyclic shift of the row by G r elements = G r / 8 of the 16-byte pieces the MFMA operands are made of, i.e. a
//     rolled row is staged by reading every piece from a shifted place -- as long as all the columns of a tile
//     belong to ONE reference track, so that a row has one roll: the chroma tile list breaks the columns at track
//     ends (W.tx = first column group, W.pad = groups in the tile; mfcc / ssm tiles run across tracks).
// ------------------------------------------------------------------------------------
constexpr int EFR_ROWS = 256, EFR_COLS = 128, EFR_THREADS = 512;
constexpr int EFR_A = 3 * EFR_ROWS * EFB_LP, EFR_B = 3 * EFR_COLS * EFB_LP;      // bf16 elements of one buffer
constexpr int EFR_TP = 32;                                                       // pitch (floats) of a wave's turning tile in the epilogue
constexpr int EFR_LDS_BYTES = 2 * 2 * (EFR_A + EFR_B) + 8 * 16 * EFR_TP * 4;     // 147 456 + 16 384 = all 160 KB

#ifdef ACX_EF_TIMING   /* development builds (scripts/ab_build_acx.sh timing -DACX_EF_TIMING; scripts/ef_phase_timing.py): where a tile's time goes */
__device__ unsigned long long g_ef_clk[16];          // [0] start-up, [1] k loop, [2] epilogue until the last store is issued, [3] until it is acknowledged, [15] waves
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// F16 = 1 (ACX_EF_GEMM_F16X2): the pool holds TWO fp16 terms of x / inv[row] (ef_split_bf16_kernel with row scales) and
// a cell takes THREE v_mfma_f32_16x16x32_f16 per chunk -- x1 y2, x2 y1, x1 y1 (x2 y2 stays below the accumulator's
// rounding: ACX_EF_F16_PRODUCTS) -- instead of six bf16 ones: 22 significant bits of every value instead of all 24,
// half of the MFMAs, two thirds of the LDS stores and pool reads.  The epilogue
// multiplies a product by inv[row] inv[column] (powers of two: exact): inv0 / inv1 = the row scales of split0 / split1.
template <int CH, int F16 = 0>
__global__ __launch_bounds__(EFR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ef_gemm_rect_bf16x3_kernel(
    const unsigned short *__restrict__ split0, const unsigned short *__restrict__ split1, const float *__restrict__ nrm0,
    const float *__restrict__ nrm1, const EfPair *__restrict__ pd, const EfSegRect *__restrict__ rects,
    const EfSegWg *__restrict__ wgs, const EfSegGroup *__restrict__ rowg, const EfSegGroup *__restrict__ colg, const int32_t *__restrict__ pairtab,
    float *__restrict__ scratch, int Kp0, int Kp1, const float *__restrict__ inv0, const float *__restrict__ inv1)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short efr_lds[];
    unsigned short *As = efr_lds;                    // [buffer][term][row][32 k]
    unsigned short *Bs = efr_lds + 2 * EFR_A;
#ifdef ACX_EF_TIMING
    const unsigned long long clk0_ = __builtin_readcyclecounter();
#endif
    const EfSegWg W = wgs[blockIdx.x];               // ty in units of 16 row groups; tx = first column group, pad = column groups
    const EfSegRect R = rects[W.rect];
    const int ty = W.ty, tx = W.tx, ncg = W.pad;
    const int s = CH ? 2 : (int)blockIdx.z;          // 0 mfcc, 1 ssm, 2 chroma
    const int Kp = (CH || s == 0) ? Kp0 : Kp1;
    const unsigned short *S = (CH || s == 0) ? split0 : split1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int NA = 4, NB = 4;
    constexpr int NT = F16 ? 2 : 3;                  // terms of a value in the pool: a row's 32-k chunk is NT x 64 contiguous bytes
    const int gr0 = 16 * ty + NA * wr, gc0 = tx + NB * wc;
    // the wave's 4 row and 4 column groups, and the pair of every sub-tile (-1: nothing to store) -- wave-uniform.
    // Straight-line loads with clamped indices (8 group records, then 16 table entries: two latencies, not 32)
    EfSegGroup GA[NA], GB[NB];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const bool in = gr0 + a < R.ng;
        GA[a] = rowg[R.g0 + (in ? gr0 + a : 0)];
        if (!in) GA[a].valid = 0;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const bool in = NB * wc + b < ncg;
        GB[b] = colg[R.h0 + (in ? gc0 + b : tx)];
        if (!in) GB[b].valid = 0;
    }
    int pidx[NA][NB];
    bool any = false;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            int p = pairtab[R.ptab0 + GA[a].slot * R.ncols + GB[b].slot];
            p = (GA[a].valid > 0 && GB[b].valid > 0) ? p : -1;
            pidx[a][b] = __builtin_amdgcn_readfirstlane(p);
            any = any || p >= 0;
        }
#ifdef ACX_EF_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long clkA_ = __builtin_readcyclecounter();     // the wave's groups and pairs are known
#endif
    f32x4 acc[NA][NB];
    float zero_ = 0.0f;
    if (F16) asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));     // (a register, not the inline constant: band_kernel's note on this MFMA; scripts/isa_lint.py checks what the compiler made of the chain)
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{zero_, zero_, zero_, zero_};

    // staging: thread -> 16-byte piece tid % 4 of rows tid / 4 and 128 + tid / 4 of A, row tid / 4 of B, every term.
    // Rows that do not exist (behind a track's last block, behind the rectangle) read pool row 0: their products only
    // reach cells that are never stored, and the loads stay unconditional.
    const int srow = tid >> 2, sp = tid & 3;
    const int sg = wave, sr = srow & 15;               // (srow / 16 == the wave's number: its group records arrive by scalar loads,
                                                       //  in the same batch as GA / GB above)
    const int pieces = Kp / 8;                         // 16-byte pieces of a row (per term)
    const unsigned short *ap0 = S, *ap1 = S, *bp = S + sp * 8;
    int sp0 = sp, sp1 = sp;                            // (CH) the piece of the source row that lands at piece tid % 4 of the chunk
    {
        int rslot = 0;
        if (CH) rslot = colg[R.h0 + tx].slot;          // the tile's one reference track
        auto roll_of = [&](const EfSegGroup &g) {
            const int p = pairtab[R.ptab0 + g.slot * R.ncols + rslot];
            int r = p >= 0 ? pd[p].oti : 0;
            r = (sp - (pieces / 12) * r) % pieces;     // piece - G r / 8, into [0, pieces)
            return r < 0 ? r + pieces : r;
        };
        if (16 * ty + sg < R.ng) {
            const EfSegGroup g = rowg[R.g0 + 16 * ty + sg];
            if (sr < g.valid) { ap0 = S + (g.poolrow + sr) * NT * Kp; if (CH) sp0 = roll_of(g); }
        }
        if (16 * ty + 8 + sg < R.ng) {
            const EfSegGroup g = rowg[R.g0 + 16 * ty + 8 + sg];
            if (sr < g.valid) { ap1 = S + (g.poolrow + sr) * NT * Kp; if (CH) sp1 = roll_of(g); }
        }
        if (sg < ncg) {
            const EfSegGroup g = colg[R.h0 + tx + sg];
            if (sr < g.valid) bp = S + (g.poolrow + sr) * NT * Kp + sp * 8;
        }
        if (!CH) { ap0 += sp * 8; ap1 += sp * 8; }
    }
#ifdef ACX_EF_TIMING
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long clkB_ = __builtin_readcyclecounter();     // the staging rows are known
#endif
    u32x4 st[9];                                      // pieces 0-2: A rows tid / 4, 3-5: A rows 128 + tid / 4, 6-8: B (one per term)
    auto gload_piece = [&](auto p_tag) {
        constexpr int p = decltype(p_tag)::value;
        if (F16 && p % 3 == 2) return;                 // (the third term does not exist)
        const unsigned short *src;
        if (p >= 6) src = bp;
        else if (!CH) src = p < 3 ? ap0 : ap1;
        else {                                         // [k / 32][term][k % 32]: piece q of the row sits at 96 (q / 4) + 8 (q % 4)
            const int q = p < 3 ? sp0 : sp1;
            src = (p < 3 ? ap0 : ap1) + (32 * NT) * (q >> 2) + 8 * (q & 3);
        }
        st[p] = *reinterpret_cast<const u32x4 *>(src + (p % 3) * EFB_BK);
    };
    auto gload_advance = [&]() {
        bp += NT * EFB_BK;
        if (!CH) { ap0 += NT * EFB_BK; ap1 += NT * EFB_BK; }
        else {
            sp0 += 4; sp0 = sp0 >= pieces ? sp0 - pieces : sp0;
            sp1 += 4; sp1 = sp1 >= pieces ? sp1 - pieces : sp1;
        }
    };
    // LDS rows are 64 bytes (32 k of one term); the four 16-byte pieces of a row sit at piece ^ swz(row / 4): a
    // ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH, LDS),
    // i.e. rows 0-3 and 12-15 at one k piece together with rows 4-11 at the next: swz = 0, 2, 3, 1 for row / 4 = 0 .. 3 puts
    // the 16 lanes of every group on 16 different bank quads (piece ^ (row / 4), E1b's choice, leaves them 2-way
    // conflicted: a third of that kernel's LDS cycles); the stores (the 4 pieces of a row by 4 neighbouring lanes) are
    // conflict free under any such permutation.
    const int skl = (sp ^ ((0x78 >> (2 * ((srow >> 2) & 3))) & 3)) * 8;
    unsigned short *as0 = As + srow * EFB_LP + skl, *bs0 = Bs + srow * EFB_LP + skl;
    auto lstore_piece = [&](int buf, auto p_tag) {
        constexpr int p = decltype(p_tag)::value;
        if (F16 && p % 3 == 2) return;
        unsigned short *dst = p < 3 ? as0 + buf * EFR_A + (p * EFR_ROWS) * EFB_LP
                            : (p < 6 ? as0 + buf * EFR_A + ((p - 3) * EFR_ROWS + 128) * EFB_LP : bs0 + buf * EFR_B + ((p - 6) * EFR_COLS) * EFB_LP);
#ifdef ACX_EF_NO_LSTORE   /* ablation build (scripts/ab_build_acx.sh nolst -DACX_EF_NO_LSTORE; WRONG matrices): what the k loop costs without its ds_write pass --
                             the ceiling of staging by LDS-DMA (global_load_lds_dwordx4), profiles/r05_ef.md (b) */
        { const u32x4 keep_ = st[p]; unsigned short *const d_ = dst; asm volatile("" :: "v"(keep_), "v"(d_)); }
#else
        *reinterpret_cast<u32x4 *>(dst) = st[p];
#endif
    };
    auto for9 = [&](auto &&f) {
        f(std::integral_constant<int, 0>()); f(std::integral_constant<int, 1>()); f(std::integral_constant<int, 2>());
        f(std::integral_constant<int, 3>()); f(std::integral_constant<int, 4>()); f(std::integral_constant<int, 5>());
        f(std::integral_constant<int, 6>()); f(std::integral_constant<int, 7>()); f(std::integral_constant<int, 8>());
    };
    const int lks = lk ^ ((0x78 >> (2 * ((lr >> 2) & 3))) & 3);
    const unsigned short *aop = As + (64 * wr + lr) * EFB_LP + 8 * lks;
    const unsigned short *bop = Bs + (64 * wc + lr) * EFB_LP + 8 * lks;

    // ---- one k chunk: 24 groups of 4 MFMAs (one term pair x the 4 row sub-tiles); behind every other group one staging
    // piece goes to LDS (chunk k + 1) and is loaded again (chunk k + 2); behind the second group of a column sub-tile the
    // operands of the next one are read.  Sub-tiles without a pair are multiplied like the others (never stored).
    // The workgroup's ONE barrier of a chunk stands behind group 18 (the last LDS store of chunk k + 1); behind it the
    // operands of the first two groups of chunk k + 1 are read into registers of their own while groups 19 - 23 run, so
    // that the matrix pipe does not drain at the chunk boundary (8 waves x 15 reads after a barrier at the boundary
    // kept it idle for ~10 % of a chunk).
    constexpr int TP = F16 ? 1 : 2;                    // the term the chunk's first products take (F16: that is all of them)
    bf16x8 pa0[NA], pa2[NA], pb0, pb2;                 // prefetched: av[.][0], av[.][TP], bv[0][0], bv[0][TP] of the next chunk
    auto prefetch = [&](int buf) {
        const unsigned short *a_ = aop + buf * EFR_A, *b_ = bop + buf * EFR_B;
        pb2 = *reinterpret_cast<const bf16x8 *>(b_ + (TP * EFR_COLS) * EFB_LP);
#pragma unroll
        for (int a = 0; a < NA; ++a) pa0[a] = *reinterpret_cast<const bf16x8 *>(a_ + (16 * a) * EFB_LP);
        pb0 = *reinterpret_cast<const bf16x8 *>(b_);
#pragma unroll
        for (int a = 0; a < NA; ++a) pa2[a] = *reinterpret_cast<const bf16x8 *>(a_ + (TP * EFR_ROWS + 16 * a) * EFB_LP);
    };
    auto chunk_mma = [&](int cur, auto st_tag, auto ld_tag) {
        constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
#ifndef ACX_EF_F16_PRODUCTS
#define ACX_EF_F16_PRODUCTS 3      /* 3: x1 y2 + x2 y1 + x1 y1.  x2 y2 is <= 2^-22 |x y| per element -- a chunk's 32 such terms almost never reach the
                                      accumulator's last bit: with it (4; the round's first version) the 500-track parity set gives the SAME moved-score
                                      counts and MAPs to every printed digit, and the worst d^2 error against f64 moves from 7.5e-7 to 8.0e-7 of the
                                      scale (bar 4e-6); without it the GEMMs take 21.5 instead of 24.7 ms per 8128 pairs */
#endif
        constexpr bool P3 = F16 && ACX_EF_F16_PRODUCTS == 3;
        constexpr int NG = F16 ? (P3 ? 3 : 4) : 6;                                   // products of a cell and chunk
        constexpr int TA[6] = {F16 ? (P3 ? 0 : 1) : 0, F16 ? (P3 ? 1 : 0) : 2, P3 ? 0 : 1, 0, 1, 0},
                      TB[6] = {F16 ? 1 : 2, F16 ? (P3 ? 0 : 1) : 0, F16 ? 0 : 1, F16 ? 0 : 1, 0, 0};   // smallest products first, as in E1b
        constexpr int LASTP = F16 ? (P3 ? 8 : 11) : 18;                              // the slot of the chunk's barrier (behind its last staging piece and its last operand read)
        const unsigned short *a_ = aop + cur * EFR_A, *b_ = bop + cur * EFR_B;
        bf16x8 av[NA][3], bv[2][3];
#ifdef ACX_EF_ABL_NOREAD       /* ... without the operand reads behind the chunk's first (all MFMAs on the prefetched registers) */
        auto rdb = [&](int e, int b, int q) { bv[e][q] = bv[0][q]; (void)b; };
#else
        auto rdb = [&](int e, int b, int q) { bv[e][q] = *reinterpret_cast<const bf16x8 *>(b_ + (q * EFR_COLS + 16 * b) * EFB_LP); };
#endif
#pragma unroll
        for (int a = 0; a < NA; ++a) { av[a][0] = pa0[a]; av[a][TP] = pa2[a]; }
        bv[0][TP] = pb2; bv[0][0] = pb0;
        if (!F16) {
            rdb(0, 0, 1);
#pragma unroll
            for (int a = 0; a < NA; ++a) av[a][1] = *reinterpret_cast<const bf16x8 *>(a_ + (EFR_ROWS + 16 * a) * EFB_LP);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    if (F16) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bv[b & 1][TB[g]]), __builtin_bit_cast(f16x8, av[a][TA[g]]), acc[a][b], 0, 0, 0);
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[b & 1][TB[g]], av[a][TA[g]], acc[a][b], 0, 0, 0);
                }
                if (g == 1 && b + 1 < NB) {
                    if (F16) { rdb((b + 1) & 1, b + 1, 1); rdb((b + 1) & 1, b + 1, 0); }
                    else { rdb((b + 1) & 1, b + 1, 2); rdb((b + 1) & 1, b + 1, 0); rdb((b + 1) & 1, b + 1, 1); }
                }
                const int slot = NG * b + g;
                // bf16x3: nine pieces behind slots 2, 4 .. 18; f16x2: its six behind slots 1, 3 .. 11
                if (P3 ? (slot >= 1 && slot <= 6) : (slot >= (F16 ? 1 : 2) && slot <= LASTP && (slot & 1) == (F16 ? 1 : 0))) {
                    auto piece = [&](auto p_tag) {
                        if (ST) lstore_piece(cur ^ 1, p_tag);
                        if (LD) gload_piece(p_tag);
                    };
                    const int nth = P3 ? slot - 1 : (slot - (F16 ? 1 : 2)) / 2;
                    switch (F16 ? nth + nth / 2 : nth) {                             // (f16x2: 0 1 3 4 6 7)
                    case 0: piece(std::integral_constant<int, 0>()); break;
                    case 1: piece(std::integral_constant<int, 1>()); break;
                    case 2: piece(std::integral_constant<int, 2>()); break;
                    case 3: piece(std::integral_constant<int, 3>()); break;
                    case 4: piece(std::integral_constant<int, 4>()); break;
                    case 5: piece(std::integral_constant<int, 5>()); break;
                    case 6: piece(std::integral_constant<int, 6>()); break;
                    case 7: piece(std::integral_constant<int, 7>()); break;
                    default: piece(std::integral_constant<int, 8>()); break;
                    }
                }
                if (ST && slot == LASTP) {
#ifndef ACX_EF_ABL_NOBARRIER   /* k-loop ablations (WRONG matrices; scripts/ab_build_acx.sh, profiles/r06_ef.md): the chunk without its barrier */
                    __syncthreads();
#endif
                    prefetch(cur ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (LD) gload_advance();
    };
    // ---- a wave without a pair (beside the rectangle, or beside the track of a chroma tile) only stages its share
    auto chunk_idle = [&](int cur, auto st_tag, auto ld_tag) {
        constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
        if (ST) for9([&](auto p_tag) { lstore_piece(cur ^ 1, p_tag); });
        if (LD) { for9([&](auto p_tag) { gload_piece(p_tag); }); gload_advance(); }
        if (ST) __syncthreads();
    };
    const int nk = Kp / EFB_BK;
    // prologue: chunk 0 into buffer 0, chunk 1 into the registers
    for9([&](auto p_tag) { gload_piece(p_tag); });
    gload_advance();
#ifdef ACX_EF_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long clkC_ = __builtin_readcyclecounter();     // the first chunk has arrived
#endif
    for9([&](auto p_tag) { lstore_piece(0, p_tag); });
    if (nk > 1) { for9([&](auto p_tag) { gload_piece(p_tag); }); gload_advance(); }
    __syncthreads();
    // (two copies of the loop, the same barriers in both: one loop with a per-chunk choice keeps two sets of accumulators)
    auto sweep = [&](auto &&chunk) {
        int kc = 0;
        for (; kc + 2 < nk; ++kc) chunk(kc & 1, std::true_type(), std::true_type());
        if (kc + 1 < nk) {
            chunk(kc & 1, std::true_type(), std::false_type());
            ++kc;
        }
        chunk(kc & 1, std::false_type(), std::false_type());
    };
#ifdef ACX_EF_TIMING
    const unsigned long long clk1_ = __builtin_readcyclecounter();
#endif
    if (any) {
        prefetch(0);
        sweep(chunk_mma);
    } else sweep(chunk_idle);
#ifdef ACX_EF_TIMING
    const unsigned long long clk2_ = __builtin_readcyclecounter();
#endif

    // ---- epilogue: every sub-tile into the matrix of its own pair (get_csm: sqrt(max(0, |x|^2 + |y|^2 - 2 x.y));
    // get_csm_cosine of unit rows: 1 - x.y).  The reference blocks are the MFMA's ROW operand, so a lane's four
    // accumulator values are four consecutive COLUMNS of one row of C: one 16-byte store per sub-tile.  The transposed
    // matrices are not written (the column statistics come from C, ef_colstat_kernel) unless the pair keeps them.
    // With one workgroup per CU nothing hides this tail: the norms (8 loads per lane) and the 16 pair records are
    // fetched in two batches before the first store instead of one dependent chain per sub-tile (measured in the
    // round's first version of this epilogue: ~2 k cycles x 16 sub-tiles of a 180 k-cycle tile).
    if (!any) return;
#ifdef ACX_EF_NOEPI   /* ablation build (scripts/ab_build_acx.sh noepi -DACX_EF_NOEPI): what a tile costs without its epilogue -- 20.8 instead of 24.1 ms */
    {
        float keep = 0.0f;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) keep += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (keep == 1.2345e-33f) scratch[0] = keep;
        return;
    }
#endif
    const float *nrm = s == 0 ? nrm0 : nrm1;
    const float *inv = (CH || s == 0) ? inv0 : inv1;
    const int il = lr, jl = 4 * lk;                      // accumulator layout: row il, columns jl .. jl + 3 of the sub-tile
    float nx[NA];
    f32x4 ny[NB];
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    if (!CH) {                                           // (rows behind a group's last block read the next track's norms
#pragma unroll                                           //  or the slack behind the pool; their cells are not stored)
        for (int a = 0; a < NA; ++a) nx[a] = nrm[GA[a].poolrow + il];
#pragma unroll
        for (int b = 0; b < NB; ++b) ny[b] = *reinterpret_cast<const f32x4u *>(nrm + GB[b].poolrow + jl);
    }
    float sx[NA];
    f32x4 sy[NB];
    if (F16) {
#pragma unroll
        for (int a = 0; a < NA; ++a) sx[a] = inv[GA[a].poolrow + il];
#pragma unroll
        for (int b = 0; b < NB; ++b) sy[b] = *reinterpret_cast<const f32x4u *>(inv + GB[b].poolrow + jl);
    }
    int64_t cbase[NA][NB];                               // float offset of the sub-tile's first cell in its pair's matrix
    int cpitch[NA][NB], ctn[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const EfPair *P = pd + (pidx[a][b] < 0 ? 0 : pidx[a][b]);
            const int pc = P->pitchC;
            cbase[a][b] = P->offC + (int64_t)s * P->M * pc + (int64_t)GA[a].local0 * pc + GB[b].local0;
            cpitch[a][b] = pc;
            ctn[a][b] = P->ctN;
        }
    // A CU retires stores by the cache LINE (measured: ~140 cycles per store instruction that touches 16 lines, whatever
    // it fills of them -- 7 B / cycle / CU; the tail of 128 such instructions per workgroup was a sixth of a tile's
    // time).  A sub-tile's accumulators are 16 rows x 64 bytes: two neighbouring sub-tiles of the same pair are turned
    // round through a wave-private LDS tile (16 rows x 32 columns, in the 16 KB of LDS the operand buffers leave free)
    // so that an instruction writes 8 rows x 128 bytes = 8 FULL lines; non-temporal (the matrices are read again by
    // the next kernels, long after the L2 has turned over).
    float *Tw = reinterpret_cast<float *>(efr_lds + 2 * (EFR_A + EFR_B)) + wave * (16 * EFR_TP);
    auto value = [&](int a, int b, float (&v)[4]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            // (two multiplications: the product of two scales may leave f32's range where the rescaled sum does not)
            const float dot = F16 ? (acc[a][b][reg] * sx[a]) * sy[b][reg] : acc[a][b][reg];
            if (CH) v[reg] = 1.0f - dot;
            else {
                float tq = (nx[a] + ny[b][reg]) - 2.0f * dot;
                if (tq < 0.0f) tq = 0.0f;
                v[reg] = ef_sqrt_nonneg(tq);
            }
        }
    };
    auto narrow = [&](int a, int b) {                                      // one sub-tile by itself (rims, track ends)
        if (pidx[a][b] < 0) return;                                        // wave-uniform
        float v[4];
        value(a, b, v);
        float *cr = scratch + cbase[a][b] + (int64_t)il * cpitch[a][b] + jl;
        if (GA[a].valid == 16 && GB[b].valid == 16) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(cr));
        else if (il < GA[a].valid) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (jl + reg < GB[b].valid) cr[reg] = v[reg];
        }
        if (ctn[a][b] && il < GA[a].valid) {                               // (K > EF_COLSTAT_MAXK: rare, narrow stores)
            const EfPair P = pd[pidx[a][b]];
            float *ct = scratch + ef_ct_off(P, s) + (size_t)(GB[b].local0 + jl) * P.pitchT + GA[a].local0 + il;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (jl + reg < GB[b].valid) ct[(size_t)reg * P.pitchT] = v[reg];
        }
    };
    const int tr = lane >> 3, tc = 4 * (lane & 7);       // after the turn: rows tr and 8 + tr, columns tc .. tc + 3 of 32
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; b += 2) {
            const bool wide = pidx[a][b] >= 0 && pidx[a][b] == pidx[a][b + 1] && GA[a].valid == 16 && GB[b].valid == 16 &&
                              GB[b + 1].valid == 16 && GB[b + 1].local0 == GB[b].local0 + 16 && !ctn[a][b];
            if (!wide) { narrow(a, b); narrow(a, b + 1); continue; }       // wave-uniform
            float v0[4], v1[4];
            value(a, b, v0);
            value(a, b + 1, v1);
            // (the eight 16-byte pieces of a tile row sit at piece ^ (row / 2): conflict free both ways)
            const int wz = (il >> 1) & 7;
            *reinterpret_cast<float4 *>(Tw + il * EFR_TP + 4 * (lk ^ wz)) = make_float4(v0[0], v0[1], v0[2], v0[3]);
            *reinterpret_cast<float4 *>(Tw + il * EFR_TP + 4 * ((4 + lk) ^ wz)) = make_float4(v1[0], v1[1], v1[2], v1[3]);
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(Tw + tr * EFR_TP + 4 * ((lane & 7) ^ ((tr >> 1) & 7)));
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(Tw + (8 + tr) * EFR_TP + 4 * ((lane & 7) ^ (((8 + tr) >> 1) & 7)));
            float *cr = scratch + cbase[a][b] + (int64_t)tr * cpitch[a][b] + tc;
            __builtin_nontemporal_store(w0, reinterpret_cast<f32x4 *>(cr));
            __builtin_nontemporal_store(w1, reinterpret_cast<f32x4 *>(cr + (int64_t)8 * cpitch[a][b]));
        }
#ifdef ACX_EF_TIMING
    const unsigned long long clk3_ = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long clk4_ = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_ef_clk[0], clk1_ - clk0_); atomicAdd(&g_ef_clk[1], clk2_ - clk1_); atomicAdd(&g_ef_clk[2], clk3_ - clk2_);
        atomicAdd(&g_ef_clk[3], clk4_ - clk3_); atomicAdd(&g_ef_clk[15], 1ull);
        atomicAdd(&g_ef_clk[5], clkA_ - clk0_); atomicAdd(&g_ef_clk[6], clkB_ - clkA_); atomicAdd(&g_ef_clk[7], clkC_ - clkB_); atomicAdd(&g_ef_clk[8], clk1_ - clkC_);
    }
#endif
}

// ------------------------------------------------------------------------------------
// E1d: the chroma (cosine) cross-similarity matrices over the same rectangles, f32 MFMA.  The blocked-OTI roll of
// the first song's bins (get_csm_blocked_oti, cross_recurrence.py:105-134) depends on the PAIR, so it cannot be
// applied while the operands are staged (a tile of a rectangle serves several pairs): the operands go to LDS
// unrolled and every lane READS its A value from the rolled k-row instead -- A1[12 g + c] = A[12 g + (c - oti) mod 12]
// -- with the roll of the wave's pair (one pair per wave in all but the few waves that straddle a track boundary;
// those read A once per sub-tile).  Same k steps, same operand values per cell as ef_gemm_kernel: same bits.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ef_gemm_seg_f32_kernel(const float *__restrict__ feat, const EfPair *__restrict__ pd,
                                                              const EfSegRect *__restrict__ rects, const EfSegWg *__restrict__ wgs,
                                                              const EfSegGroup *__restrict__ rowg, const EfSegGroup *__restrict__ colg,
                                                              const int32_t *__restrict__ pairtab, float *__restrict__ scratch, int K)
{
    __shared__ float As[EF_BK * EF_LP];
    __shared__ float Bs[EF_BK * EF_LP];
    const EfSegWg W = wgs[blockIdx.x];
    const EfSegRect R = rects[W.rect];
    const int ty = W.ty, tx = W.tx;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    const int gr0 = 8 * ty + 4 * wr, gc0 = 8 * tx + 4 * wc;
    int pidx[4][4], rot[4][4];
    bool any = false, uniform = true;
    int rot0 = -1;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            int p = -1, r = 0;
            if (gr0 + a < R.ng && gc0 + b < R.nh) {
                const EfSegGroup ga = rowg[R.g0 + gr0 + a], gb = colg[R.h0 + gc0 + b];
                if (ga.valid > 0 && gb.valid > 0) p = pairtab[R.ptab0 + ga.slot * R.ncols + gb.slot];
            }
            if (p >= 0) r = pd[p].oti;
            p = __builtin_amdgcn_readfirstlane(p);
            r = __builtin_amdgcn_readfirstlane(r);
            pidx[a][b] = p; rot[a][b] = r;
            any = any || p >= 0;
            if (p < 0) uniform = false;
            else if (rot0 < 0) rot0 = r;
            else if (r != rot0) uniform = false;
        }

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: thread -> (row = tid / 2, 12 consecutive k = 12 * (tid % 2) ...) as three float4 per operand
    const int srow = tid >> 1, sk = (tid & 1) * 12;
    const int sg = srow >> 4, sr = srow & 15;
    bool rowa = false, rowb = false;
    const float *ap = feat, *bp = feat;
    if (8 * ty + sg < R.ng) {
        const EfSegGroup g = rowg[R.g0 + 8 * ty + sg];
        rowa = sr < g.valid;
        if (rowa) ap = feat + (g.poolrow + sr) * K + sk;
    }
    if (8 * tx + sg < R.nh) {
        const EfSegGroup g = colg[R.h0 + 8 * tx + sg];
        rowb = sr < g.valid;
        if (rowb) bp = feat + (g.poolrow + sr) * K + sk;
    }
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    float ra[12], rb[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) { ra[e] = 0.f; rb[e] = 0.f; }
    auto gload = [&](int k0) {
        if (k0 + EF_BK <= K) {                       // workgroup-uniform
            if (rowa) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const f32x4 va = *reinterpret_cast<const f32x4u *>(ap + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) ra[4 * q + e] = va[e];
                }
            }
            if (rowb) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const f32x4 vb = *reinterpret_cast<const f32x4u *>(bp + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[4 * q + e] = vb[e];
                }
            }
            ap += EF_BK;
            bp += EF_BK;
        } else {                                     // the block that crosses K: element-wise, zeros beyond K
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const bool ok = k0 + sk + e < K;
                ra[e] = (ok && rowa) ? ap[e] : 0.f;
                rb[e] = (ok && rowb) ? bp[e] : 0.f;
            }
        }
    };
    const int swrow = srow ^ (16 * (tid & 1));       // (the XOR of the second 12-group, as in ef_gemm_kernel)
    float *as0 = As + sk * EF_LP + swrow, *bs0 = Bs + sk * EF_LP + swrow;
    auto lstore = [&]() {
#pragma unroll
        for (int e = 0; e < 12; ++e) { as0[e * EF_LP] = ra[e]; bs0[e * EF_LP] = rb[e]; }
    };
    // rolled k-row of A for k-step kb of the lane: c = (4 kb + lk) mod 12 = 4 (kb mod 3) + lk
    auto arow = [&](int kb, int r) {
        int idx = 4 * (kb % 3) + lk - r;
        idx = idx < 0 ? idx + 12 : idx;
        return 12 * (kb / 3) + idx;
    };
    int arow_u[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { int idx = 4 * q + lk - (rot0 < 0 ? 0 : rot0); arow_u[q] = idx < 0 ? idx + 12 : idx; }
    gload(0);
    for (int k0 = 0; k0 < K; k0 += EF_BK) {
        lstore();
        __syncthreads();
        if (k0 + EF_BK < K) gload(k0 + EF_BK);           // in flight during the MFMAs below
        if (uniform) {
#pragma unroll
            for (int kb = 0; kb < EF_BK / 4; ++kb) {
                const int sw = 16 * (kb / 3);
                float av[4], bv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) av[a] = As[(12 * (kb / 3) + arow_u[kb % 3]) * EF_LP + ((64 * wr + 16 * a + lr) ^ sw)];
#pragma unroll
                for (int b = 0; b < 4; ++b) bv[b] = Bs[(4 * kb + lk) * EF_LP + ((64 * wc + 16 * b + lr) ^ sw)];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
        } else if (any) {
#pragma unroll
            for (int kb = 0; kb < EF_BK / 4; ++kb) {
                const int sw = 16 * (kb / 3);
                float bv[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) bv[b] = Bs[(4 * kb + lk) * EF_LP + ((64 * wc + 16 * b + lr) ^ sw)];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (pidx[a][b] >= 0) {
                            const float ava = As[arow(kb, rot[a][b]) * EF_LP + ((64 * wr + 16 * a + lr) ^ sw)];
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ava, bv[b], acc[a][b], 0, 0, 0);
                        }
            }
        }
        __syncthreads();
    }
    // ---- epilogue: get_csm_cosine (rows are unit vectors: 1 - dot), every sub-tile into its own pair's matrices
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (pidx[a][b] < 0) continue;
            const EfPair P = pd[pidx[a][b]];
            const EfSegGroup ga = rowg[R.g0 + gr0 + a], gb = colg[R.h0 + gc0 + b];
            float *C = scratch + ef_c_off(P, 2);
            float *CT = scratch + ef_ct_off(P, 2);
            const int il = 4 * lk, jl = lr;
            const int ib = ga.local0 + il, j = gb.local0 + jl;
            const bool jok = jl < gb.valid;
            float v[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                v[reg] = 1.0f - acc[a][b][reg];
                if (il + reg < ga.valid && jok) C[(size_t)(ib + reg) * P.pitchC + j] = v[reg];
            }
            if (jok && P.ctN) {
                float *ct = CT + (size_t)j * P.pitchT + ib;
                if (il + 3 < ga.valid) *reinterpret_cast<float4 *>(ct) = make_float4(v[0], v[1], v[2], v[3]);
                else
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        if (il + reg < ga.valid) ct[reg] = v[reg];
            }
        }
}

// ------------------------------------------------------------------------------------
// E2: per-row statistics of a matrix with rows <= 256 NQ long (NQ = 2: 512, NQ = 4: 1024).  mode 0 (C rows): threshold
// t_i = the kb-th smallest (k = round(kappa n), kappa < 1; kappa >= 1: k = kappa; k = 0 or
// kappa == 0 handled by the host) and r_i = mean of the kw smallest; mode 1 (C^T rows):
// only the mean (= column statistic c_j); mode 2 (F rows): only the threshold.
// ------------------------------------------------------------------------------------
// lane exchanges of the row statistics by DPP (no LDS round trip; __shfl_xor is a ds_bpermute_b32: ~100 cycles each, six of them
// in a row's sum -- the neighbourhood mean cost more than the selection in front of it): xor 1 / xor 2 inside a quad, then the
// mirrored half row / row -- the partner differs from xor 4 / xor 8, but after the quad steps every lane of a quad (of a half
// row) holds the same partial value, so the result is the xor butterfly's.
template <int CTRL> __device__ __forceinline__ int ef_dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float ef_dpp_f(float v) { return __int_as_float(ef_dpp_i<CTRL>(__float_as_int(v))); }
constexpr int EF_DPP_XOR1 = 0xB1, EF_DPP_XOR2 = 0x4E, EF_DPP_HALF_MIRROR = 0x141, EF_DPP_ROW_MIRROR = 0x140;
// sum over the wave, the same in every lane: a balanced tree (lanes 1, 2, 4, 8 apart, then the four rows as (r0 + r1) + (r2 + r3))
__device__ __forceinline__ float ef_wave_sum_f(float s)
{
    s += ef_dpp_f<EF_DPP_XOR1>(s);
    s += ef_dpp_f<EF_DPP_XOR2>(s);
    s += ef_dpp_f<EF_DPP_HALF_MIRROR>(s);
    s += ef_dpp_f<EF_DPP_ROW_MIRROR>(s);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 48));
    return (r0 + r1) + (r2 + r3);
}

template <int NX>
__device__ __forceinline__ float mean_k_smallest(const float (&x)[NX], int kw, float vk, int lane)
{
    // vk = kw-th smallest (rank kw-1).  sum of elements < vk, plus (kw - count) * vk: exact
    // whatever the ties.
    float acc = 0.0f;
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < NX; ++t) {
        const bool lt = x[t] < vk;
        acc += lt ? x[t] : 0.0f;
        cnt += lt ? 1 : 0;
    }
    // deterministic wave sums
    const int tot = wave_sum_i(cnt);
    const float s = ef_wave_sum_f(acc);
    (void)lane;
    return (s + (float)(kw - tot) * vk) / (float)kw;
}

// Two order statistics of a row from ONE pivot-filtered histogram (wave_select_pivot, serra09_kernels.hpp, with 256 bins
// and lane pairs): the row statistics want rank K - 1 (the neighbourhood mean) and rank kappa N - 1 (the threshold) of
// the same row, both far below the pivot -- minimum, pivot estimate, binning, the masked atomics and the scan are shared,
// only the two target bins are gathered and ranked one after the other.  false: the caller selects them separately.
template <int NV>
__device__ __forceinline__ bool ef_select_pivot2(const float (&x)[NV], int k1, int k2, unsigned hist_addr, float *cand, int lane,
                                                 float &v1, float &v2, bool lane_has_data, bool group_full, float delta)
{
    constexpr int NB = 256, BPL = 4;
    const float INF = __builtin_inff();
    unsigned mnl = 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const unsigned b = __float_as_uint(x[t]);
        mnl = b < mnl ? b : mnl;
    }
    unsigned gmn = mnl;
    {
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)gmn, (int)gmn, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        gmn = o < gmn ? o : gmn;
    }
    const unsigned mnu = (unsigned)__builtin_amdgcn_readlane(wave_scan_bits((int)mnl, -1, OpMinU()), 63);
    const int mxg = __builtin_amdgcn_readlane(wave_scan_bits(group_full ? (int)gmn : (int)0x80000000, (int)0x80000000, OpMaxI()), 63);
    if (mxg < 0 || (int)mnu < 0) return false;
    const float mn = __uint_as_float(mnu);
    const float gm = __uint_as_float((unsigned)mxg);
    const float range = __builtin_fmaf(delta, gm - mn, gm) - mn;
    if (!(range >= 1e-30f) || !(range <= 1e30f) || !(mn <= 2048.0f * range)) return false;
    constexpr unsigned MAGIC = 0x4B000000u;            // 2^23: the bin comes out of the fma as MAGIC + bin (see wave_select_pivot)
    const float scale = ((float)NB - 3.0f) * __builtin_amdgcn_rcpf(range);
    const float offm = lane_has_data ? (8388609.0f - mn * scale) : INF;
    unsigned off[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) off[t] = __float_as_uint(__builtin_fmaf(x[t], scale, offm));
    {
        const unsigned nb = __builtin_amdgcn_readfirstlane(MAGIC + NB);
        const unsigned hb = __builtin_amdgcn_readfirstlane(hist_addr - 4u * MAGIC);
        unsigned one = 1u;
        asm volatile("" : "+v"(one));
        static_assert(NV % 4 == 0, "atomics go in groups of 4");
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            unsigned long long m0, m1, m2, m3, sv;
            unsigned a0, a1, a2, a3;
            asm volatile("v_cmp_gt_u32_e64 %[m0], %[nb], %[q0]\n\t"
                         "v_cmp_gt_u32_e64 %[m1], %[nb], %[q1]\n\t"
                         "v_cmp_gt_u32_e64 %[m2], %[nb], %[q2]\n\t"
                         "v_cmp_gt_u32_e64 %[m3], %[nb], %[q3]\n\t"
                         "v_lshl_add_u32 %[a0], %[q0], 2, %[hb]\n\t"
                         "v_lshl_add_u32 %[a1], %[q1], 2, %[hb]\n\t"
                         "v_lshl_add_u32 %[a2], %[q2], 2, %[hb]\n\t"
                         "v_lshl_add_u32 %[a3], %[q3], 2, %[hb]\n\t"
                         "s_mov_b64 %[sv], exec\n\t"
                         "s_mov_b64 exec, %[m0]\n\t"
                         "ds_add_u32 %[a0], %[one]\n\t"
                         "s_mov_b64 exec, %[m1]\n\t"
                         "ds_add_u32 %[a1], %[one]\n\t"
                         "s_mov_b64 exec, %[m2]\n\t"
                         "ds_add_u32 %[a2], %[one]\n\t"
                         "s_mov_b64 exec, %[m3]\n\t"
                         "ds_add_u32 %[a3], %[one]\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [sv] "=&s"(sv),
                           [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)
                         : [q0] "v"(off[t]), [q1] "v"(off[t + 1]), [q2] "v"(off[t + 2]), [q3] "v"(off[t + 3]),
                           [nb] "s"(nb), [hb] "s"(hb), [one] "v"(one)
                         : "memory");
        }
    }
    wave_lds_fence();
    // scan: lane owns bins [4 lane, +4)
    const u32x4 h = *(const lds_u32x4 *)(hist_addr + (unsigned)(lane * 4) * 4u);
    const int lsum = (int)(h.x + h.y) + (int)(h.z + h.w);
    const int incl = wave_incl_scan_i(lsum);
    const int L1 = __ffsll((long long)__ballot(incl > k1)) - 1;
    const int L2 = __ffsll((long long)__ballot(incl > k2)) - 1;
    if (L1 < 0 || L2 < 0) return false;              // fewer than k2 + 1 cells at or below the pivot
    const int ex1 = __builtin_amdgcn_readlane(incl - lsum, L1);
    const int ex2 = __builtin_amdgcn_readlane(incl - lsum, L2);
    // second level: lanes 0..3 look at lane L1's bins, lanes 16..19 at lane L2's
    const int e = lane & 15;
    const bool lo16 = lane < 16;
    int c = 0;
    if (lane < 32 && e < BPL) c = (int)*(const lds_u32 *)(hist_addr + (unsigned)((lo16 ? L1 : L2) * BPL + e) * 4u);
    int Pp = c;
    Pp += __builtin_amdgcn_update_dpp(0, Pp, 0x111, 0xf, 0xf, false);
    Pp += __builtin_amdgcn_update_dpp(0, Pp, 0x112, 0xf, 0xf, false);
    const int l1 = __ffsll((long long)__ballot(lo16 && e < BPL && ex1 + Pp > k1)) - 1;
    const int l2 = __ffsll((long long)__ballot(!lo16 && lane < 32 && e < BPL && ex2 + Pp > k2)) - 1;
    if (l1 < 0 || l2 < 0) return false;
    const int cnt1 = __builtin_amdgcn_readlane(c, l1), cnt2 = __builtin_amdgcn_readlane(c, l2);
    const int cum1 = ex1 + __builtin_amdgcn_readlane(Pp, l1) - cnt1, cum2 = ex2 + __builtin_amdgcn_readlane(Pp, l2) - cnt2;
    const int bin1 = L1 * BPL + l1, bin2 = L2 * BPL + (l2 - 16);
    if (cnt1 > 64 || cnt2 > 64) return false;
    // Round 5: most of the time no member has to be gathered at all.  The rank sits at position p = k - cum of its bin of cnt cells:
    // p == 0 -> the bin's SMALLEST cell, p == cnt - 1 -> its LARGEST (at ~0.4 cells per bin the bin holds one or two cells nine
    // times in ten).  The cells of the two target bins post their minimum / maximum with exec-masked LDS atomics on four dwords
    // (a handful of lanes; unsigned patterns order like the non-negative values) -- no candidate list, no ranking loop, no
    // readlane picks: ~70 of the kernel's 390 VALU instructions per row.
    {
        const int p1 = k1 - cum1, p2 = k2 - cum2;
        if ((p1 == 0 || p1 == cnt1 - 1) && (p2 == 0 || p2 == cnt2 - 1)) {
            lds_u32 *ext = (lds_u32 *)(unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)cand;     // [0] min1 [1] max1 [2] min2 [3] max2
            if (lane < 4) ext[lane] = (lane & 1) ? 0u : 0xFFFFFFFFu;
            wave_lds_fence();
            const unsigned a1 = MAGIC + (unsigned)bin1, a2 = MAGIC + (unsigned)bin2;
#pragma unroll
            for (int t = 0; t < NV; t += 4) {
                bool h1[4], h2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { h1[u] = off[t + u] == a1; h2[u] = off[t + u] == a2; }
                if (__ballot(h1[0] || h1[1] || h1[2] || h1[3] || h2[0] || h2[1] || h2[2] || h2[3]) != 0ull) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const unsigned xb = __float_as_uint(x[t + u]);
                        if (h1[u]) {
                            __hip_atomic_fetch_min(ext + 0, xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_max(ext + 1, xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (h2[u]) {
                            __hip_atomic_fetch_min(ext + 2, xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_max(ext + 3, xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
            wave_lds_fence();
            const u32x4 ev = *(const lds_u32x4 *)ext;
            wave_lds_fence();
            v1 = __uint_as_float(p1 == 0 ? ev.x : ev.y);
            v2 = __uint_as_float(p2 == 0 ? ev.z : ev.w);
            return true;
        }
    }
    // the members of one bin, gathered and ranked (LDS broadcasts); value of rank `want` among them
    auto resolve = [&](int bin, int ncand, int want, float &out) -> bool {
        const unsigned a1 = MAGIC + (unsigned)bin;
        int n = 0;
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            const bool h0 = off[t] == a1, h1 = off[t + 1] == a1, h2 = off[t + 2] == a1, h3 = off[t + 3] == a1;
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            if ((m0 | m1 | m2 | m3) != 0ull) {       // most groups hold no member of the target bin
                const unsigned long long mm[4] = {m0, m1, m2, m3};
                const bool hh[4] = {h0, h1, h2, h3};
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (mm[u] != 0ull) {
                        if (hh[u]) {
                            const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mm[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm[u], 0u));
                            cand[(n + (int)pos) & 63] = x[t + u];
                        }
                        n += __popcll(mm[u]);
                    }
            }
        }
        wave_lds_fence();
        if (lane >= ncand) cand[lane] = INF;
        wave_lds_fence();
        const float mine = cand[lane];
        int rank = 0;
#pragma unroll 1
        for (int t = 0; t < ncand; t += 4) {
            const float4 o = *reinterpret_cast<const float4 *>(cand + t);
            rank += (o.x < mine || (o.x == mine && t + 0 < lane)) ? 1 : 0;
            rank += (o.y < mine || (o.y == mine && t + 1 < lane)) ? 1 : 0;
            rank += (o.z < mine || (o.z == mine && t + 2 < lane)) ? 1 : 0;
            rank += (o.w < mine || (o.w == mine && t + 3 < lane)) ? 1 : 0;
        }
        const int s1 = __ffsll((long long)__ballot(lane < ncand && rank == want)) - 1;
        wave_lds_fence();
        if (s1 < 0) return false;
        out = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), s1));
        return true;
    };
    if (!resolve(bin1, cnt1, k1 - cum1, v1)) return false;
    if (!resolve(bin2, cnt2, k2 - cum2, v2)) return false;
    return true;
}

// getWCSM's kernel weight of one cell (similarity_fusion.py:15-46): exp(-C^2 / (2 (eps / 2)^2)), eps = (r_i + c_j + C) / 3, which
// is exp(-18 (C / (r_i + c_j + C))^2): one reciprocal and one exp2 instead of two IEEE divisions and an expf (the
// fused matrix is only RANKED afterwards; 1e-6 relative against the reference's f32 numpy, bound 2e-4 in the tests --
// the exact form made the fusion kernels compute bound at ~150 instructions per cell).
__device__ __forceinline__ float ef_wcsm_weight(float ri, float cj, float cv)
{
    const float q = cv * __builtin_amdgcn_rcpf((ri + cj) + cv);
    return __builtin_amdgcn_exp2f(-25.968510740383334f * (q * q));       // 18 log2(e)
}
__device__ __forceinline__ float ef_fused_value(float wsum) { return __builtin_amdgcn_exp2f(-1.4426950408889634f * wsum); }
// The same two in the reference's own operation order (acx_set_ef_fuse(ACX_EF_FUSE_EXACT); similarity_fusion.py:50-54 on float32
// arrays): Eps = (r_i + c_j + C) / 3, W = exp(-C^2 / (2 (0.5 Eps)^2)) with IEEE divisions and expf; fused = exp(-sum W).
// ~150 instructions per cell instead of ~12; differs from numpy only by the last bit of the two exponentials.
__device__ __forceinline__ float ef_wcsm_weight_exact(float ri, float cj, float cv)
{
    const float eps = ((ri + cj) + cv) / 3.0f;
    const float t = 0.5f * eps;
    return expf(-(cv * cv) / (2.0f * (t * t)));
}
__device__ __forceinline__ float ef_fused_value_exact(float wsum) { return expf(-wsum); }

#ifndef ACX_EF_ROWSTAT_WAVES
#define ACX_EF_ROWSTAT_WAVES 8      /* waves per SIMD the narrow variant (NQ = 2: rows of <= 512 cells) is compiled for: 64 registers instead of 78-80,
                                       no spills, 8 instead of 6 waves per SIMD -- the selection's chain waits on latencies: 15.3 -> 13.9 ms per 5 grid tiles */
#endif
constexpr int ef_rowstat_min_waves(int nq) { return (nq <= 2 && ACX_EF_ROWSTAT_WAVES > 0) ? ACX_EF_ROWSTAT_WAVES : 1; }
#ifdef ACX_EF_ABL   /* stage ablations of the row statistics (scripts/ab_build_acx.sh ablN -DACX_EF_ABL=N): leave after stage N, the row kept alive */
#define ACX_EF_ABL_EXIT(n_, ...) do { if (ACX_EF_ABL == (n_)) { float k_ = 0.f; for (int e_ = 0; e_ < NX; ++e_) k_ += x[e_]; const float ks_[] = {__VA_ARGS__}; for (float v_ : ks_) k_ += v_; if (k_ == 1.2345e-33f) stat[0] = k_; return; } } while (0)
#else
#define ACX_EF_ABL_EXIT(n_, ...) do { } while (0)
#endif
constexpr int EF_ROW_GB = 512;       // bins of the generic fallback (small: LDS per workgroup decides how many rows a CU works on)
// Everything the row statistics do once a wave holds ONE row in registers (x[4 q + e] = column 256 q + 4 lane + e, +inf behind the
// row): both order statistics, the neighbourhood mean, the threshold with its tie column, the binarised row.  `fhist` (256 zeroable
// dwords, 1 KB aligned), `hist`, `cand` (64 floats), `counter`: the wave's own LDS scratch.  Shared by ef_rowstat_kernel (one row per
// wave) and by the fallback of ef_rowstat2_kernel (a row its two-rows-per-wave pass could not decide).
template <int NQ, bool FUSED>
__device__ __forceinline__ void ef_row_finish(const float (&x)[4 * NQ], const EfPair &P, int s, int mode, int kw, int n, int pitch, int row,
                                              float *__restrict__ stat, unsigned *__restrict__ bits, unsigned *fhist_w, unsigned *hist_w,
                                              float *cand_w, unsigned *counter_w, int lane)
{
    constexpr int NX = 4 * NQ;
    constexpr int GB = EF_ROW_GB;
    const float INF = __builtin_inff();
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned fh_addr = (unsigned)(uintptr_t)(lds_void *)fhist_w;
    // k-th smallest (0-based) of the row: one histogram pass, generic narrowing when that cannot decide
    // Small ranks (the row-kappa threshold sits at rank ~ 0.1 n, the neighbourhood mean at rank 9) first try the
    // pivot-filtered pass of the band kernel (wave_select_pivot): only the cells below a pivot near the 0.2 quantile
    // enter the histogram, under an exec mask.  The cross-similarity values of a row crowd into a few bins (counters of
    // round 3: 69 % of this kernel's LDS cycles were bank / same-address conflicts of the unfiltered atomics); the
    // filtered pass issues an eighth of them.  It gives up (too few cells below the pivot, ties, short rows) and the
    // unfiltered pass takes over: all paths are exact.
    bool lane_has_data = false, group_full = false;        // set below
    auto kth = [&](const float (&xx)[NX], int k, int n_) -> float {
        *reinterpret_cast<uint4 *>(fhist_w + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_fence();
        float lo, hi;
        if ((k + 2) * 6 <= n_) {
            if (wave_select_pivot<NX, 256, 2>(xx, k, false, fh_addr, cand_w, lane, lo, hi, lane_has_data, group_full, 0.15f)) return lo;
            *reinterpret_cast<uint4 *>(fhist_w + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_fence();
        }
        if (wave_select_fast<NX, 256>(xx, k, false, fh_addr, cand_w, lane, lo, hi)) return lo;
        return wave_select_regs<NX, GB>(xx, k, hist_w, cand_w, counter_w, lane, false).value;
    };
    {   // which lanes / lane pairs hold cells: a pair of lanes (16 slots) takes part in the pivot estimate when at
        // least three quarters of the slots it can have in a row of this length are cells (a pair with few cells has a large
        // minimum: a pivot that filters nothing)
        int cells = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = 256 * q + 4 * lane;
            cells += j + 3 < n ? 4 : (j < n ? n - j : 0);
        }
        lane_has_data = cells > 0;
        const int cap = 8 * ((n + 255) / 256);                 // slots of a lane pair that CAN be cells in a row of n (4 per lane and 256 columns)
        group_full = cells + ef_dpp_i<EF_DPP_XOR1>(cells) >= cap - cap / 4;
    }
    ACX_EF_ABL_EXIT(1, 0.f);                             // the row (FUSED: the fused row) is in registers
    float *S = stat + P.offS + (mode >= 2 ? 3 * ef_s_stride(P) : s * ef_s_stride(P));
    // mode 0 wants two order statistics of the row: rank K - 1 (neighbourhood mean) and rank kbin - 1 (threshold).  Both
    // from one histogram when the row is long enough for the pivot filter (ef_select_pivot2), else one after the other.
    float vk2 = 0.0f, t2 = 0.0f;
    bool have2 = false;
    if (!FUSED && mode == 0) {
        const int kk = kw < n ? kw : n, kb = P.kbin;
        if (kb > kk && kb < n && (kb + 1) * 6 <= n) {
            *reinterpret_cast<uint4 *>(fhist_w + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_fence();
            have2 = ef_select_pivot2<NX>(x, kk - 1, kb - 1, fh_addr, cand_w, lane, vk2, t2, lane_has_data, group_full, 0.15f);
        }
    }
    if (!FUSED) ACX_EF_ABL_EXIT(2, vk2, t2, have2 ? 1.f : 0.f);      // + both order statistics from one histogram
    if (!FUSED && mode < 2) {                          // (before the threshold: nothing of it is alive during this selection)
        const int kk = kw < n ? kw : n;
        const float vk = have2 ? vk2 : kth(x, kk - 1, n);
        const float m = mean_k_smallest(x, kk, vk, lane);
        if (lane == 0) S[(mode == 0 ? P.pitchT : 2 * P.pitchT) + row] = m;
    }
    if (!FUSED) ACX_EF_ABL_EXIT(3, vk2, t2, have2 ? 1.f : 0.f);      // + neighbourhood mean (stored by lane 0 above)
    if (FUSED || mode != 1) {
        const int kb = P.kbin;
        float t;
        int jcut = 0x7fffffff;
        if (kb <= 0) t = -INF;                         // no neighbours: empty rows
        else if (kb >= n) t = INF;
        else {
            t = have2 ? t2 : kth(x, kb - 1, n);
            // cells equal to t: if there are more than the row may still take, find the column of the last one taken
            // (column of x[4 q + e] = 256 q + 4 lane + e).  Almost always the row has exactly kb cells <= t: one count decides
            int le = 0;
#pragma unroll
            for (int e = 0; e < NX; ++e) le += x[e] <= t ? 1 : 0;
            if (wave_sum_i(le) > kb) {                              // wave-uniform: surplus ties
                int lt = 0, eq[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) eq[q] = 0;
#pragma unroll
                for (int e = 0; e < NX; ++e) {
                    lt += x[e] < t ? 1 : 0;
                    eq[e >> 2] += x[e] == t ? 1 : 0;
                }
                const int budget = kb - wave_sum_i(lt);
                int before[NQ + 1];                                     // ties in the groups before group q
                before[0] = 0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) before[q + 1] = before[q] + wave_sum_i(eq[q]);
                int cand_j = -1;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    int incl = eq[q];
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int up = __shfl_up(incl, o, 64);
                        if (lane >= o) incl += up;
                    }
                    int rank = incl - eq[q] + before[q];           // ties before this lane's group
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (x[4 * q + e] == t) {
                            ++rank;
                            if (rank == budget) cand_j = 256 * q + 4 * lane + e;
                        }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const int other = __shfl_xor(cand_j, o, 64);
                    cand_j = cand_j > other ? cand_j : other;
                }
                jcut = cand_j;
            }
        }
        ACX_EF_ABL_EXIT(4, t, (float)jcut);              // + threshold (FUSED: its selection) and the tie count
        if (lane == 0) {
            S[row] = t;
            reinterpret_cast<int *>(S)[ef_jcut_off(P, mode >= 2 ? 3 : s) + row] = jcut;
        }
        // the binarised row (csm_to_binary: B_ij = C_ij < t_i, or C_ij == t_i and j <= jcut_i), bit j % 32 of word j / 32:
        // a lane's nibble of every 256-column group, eight lanes to a word
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            unsigned nib = 0;
            if (jcut == 0x7fffffff && t < INF) {                   // wave-uniform: every tie is taken, the pads (+inf) are not
#pragma unroll
                for (int e = 0; e < 4; ++e) nib |= x[4 * q + e] <= t ? (1u << e) : 0u;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 256 * q + 4 * lane + e;
                    const float d = x[4 * q + e];
                    nib |= ((j < n) && (d < t || (d == t && j <= jcut))) ? (1u << e) : 0u;
                }
            }
            unsigned w = nib << (4 * (lane & 7));
            w |= (unsigned)ef_dpp_i<EF_DPP_XOR1>((int)w);
            w |= (unsigned)ef_dpp_i<EF_DPP_XOR2>((int)w);
            w |= (unsigned)ef_dpp_i<EF_DPP_HALF_MIRROR>((int)w);      // (lanes 8 k .. 8 k + 7 now hold the word; lane 8 k stores it)
            const int word = 8 * q + (lane >> 3);
            if ((lane & 7) == 0 && 32 * word < pitch)
                bits[P.offB + ((int64_t)(mode >= 2 ? 3 : s) * P.M + row) * (pitch >> 5) + word] = w;
        }
    }
}

template <int NQ, bool FUSED, bool EXACT = false>
__global__ __launch_bounds__(256, ef_rowstat_min_waves(NQ)) void ef_rowstat_kernel(const EfPair *__restrict__ pd, float *__restrict__ scratch,
                                                         float *__restrict__ stat, unsigned *__restrict__ bits, int mode, int kw, int store_f)
{
    constexpr int NX = 4 * NQ;                        // values per lane
    __shared__ __attribute__((aligned(4096))) unsigned fhist[4][256];      // one-pass selection (wave_select_fast)
    __shared__ __attribute__((aligned(16))) unsigned hist[4][SelGeom<EF_ROW_GB>::SLOTS];   // generic fallback
    __shared__ __attribute__((aligned(16))) float cand[4][64];
    __shared__ unsigned counter[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const EfPair P = pd[blockIdx.y];
    const int s = blockIdx.z;                         // feature (mode 2: always 0)
    const int nrows = mode == 1 ? P.N : P.M;
    const int n = mode == 1 ? P.M : P.N;              // row length
    const int pitch = mode == 1 ? P.pitchT : P.pitchC;
    const int row = blockIdx.x * 4 + wave;
    if (row >= nrows) return;
    float x[NX];
    const float INF = __builtin_inff();
    if constexpr (!FUSED) {
        const int64_t base = mode == 0 ? ef_c_off(P, s) : (mode == 1 ? ef_ct_off(P, s) : ef_f_off(P));
        const float *v = scratch + base + (size_t)row * pitch;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = 256 * q + 4 * lane;
            float4 t = make_float4(INF, INF, INF, INF);
            if (j < pitch) t = *reinterpret_cast<const float4 *>(v + j);
            x[4 * q + 0] = (j + 0 < n) ? t.x : INF;
            x[4 * q + 1] = (j + 1 < n) ? t.y : INF;
            x[4 * q + 2] = (j + 2 < n) ? t.z : INF;
            x[4 * q + 3] = (j + 3 < n) ? t.w : INF;
        }
    } else {
        // FUSED: the row of the fused matrix, F_ij = exp(-(W0 + W1 + W2)) with getWCSM's weights (ef_wcsm_weight, the
        // arithmetic of ef_fuse_kernel to the operation), made in the registers the selection works on: the matrix is
        // binarised without a trip to memory
        float wsum[NX];
#pragma unroll
        for (int e = 0; e < NX; ++e) wsum[e] = 0.0f;
#pragma unroll
        for (int sf = 0; sf < 3; ++sf) {
            const float *Sf = stat + P.offS + sf * ef_s_stride(P);
            const float ri = Sf[P.pitchT + row];
            const float *cj = Sf + 2 * P.pitchT;
            const float *cv = scratch + ef_c_off(P, sf) + (size_t)row * pitch;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int j = 256 * q + 4 * lane;
                float4 cc = make_float4(0.f, 0.f, 0.f, 0.f), vv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < pitch) { cc = *reinterpret_cast<const float4 *>(cj + j); vv = *reinterpret_cast<const float4 *>(cv + j); }
                const float c4[4] = {cc.x, cc.y, cc.z, cc.w}, v4[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) wsum[4 * q + e] += EXACT ? ef_wcsm_weight_exact(ri, c4[e], v4[e]) : ef_wcsm_weight(ri, c4[e], v4[e]);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = 256 * q + 4 * lane;
            float f4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f4[e] = EXACT ? ef_fused_value_exact(wsum[4 * q + e]) : ef_fused_value(wsum[4 * q + e]);
                x[4 * q + e] = (j + e < n) ? f4[e] : INF;
            }
            if (store_f && j < pitch)                                      // (the debug entry point hands the matrix out)
                *reinterpret_cast<float4 *>(scratch + ef_f_off(P) + (size_t)row * pitch + j) = make_float4(f4[0], f4[1], f4[2], f4[3]);
        }
    }
    ef_row_finish<NQ, FUSED>(x, P, s, mode, kw, n, pitch, row, stat, bits, fhist[wave], hist[wave], cand[wave], &counter[wave], lane);
}

// ------------------------------------------------------------------------------------
// E2c: the column statistic c_j = mean of the kw smallest values of column j (the neighbourhood means getWCSM takes over
// the columns), from the ROWS of C: a lane owns one column and keeps its KW smallest values sorted in registers
// while the rows stream past, 64 consecutive columns per wave and load (an insertion is KW - 1 v_med3_f32 and a
// v_min_f32: new t[i] = med3(x, t[i-1], t[i])); the four waves of a workgroup take every fourth row and merge their
// lists through LDS.  Replaces the transposed matrices (a second copy of every C written by the GEMM, read once by
// a row-selection kernel) for kw <= EF_COLSTAT_MAXK; any track length.
// ------------------------------------------------------------------------------------
template <int KW>
__global__ __launch_bounds__(256) void ef_colstat_kernel(const EfPair *__restrict__ pd, const float *__restrict__ scratch,
                                                         float *__restrict__ stat, int kw)
{
    __shared__ float part[3][KW][64];
    const EfPair P = pd[blockIdx.y];
    const int s = blockIdx.z;
    if ((int)blockIdx.x * 64 >= P.N) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 64 + lane;                 // (j < pitchC: the load is inside the pair's matrix even behind column N)
    const bool ok = j < P.N;
    const float *C = scratch + ef_c_off(P, s) + j;
    const float INF = __builtin_inff();
    float t[KW];
#pragma unroll
    for (int e = 0; e < KW; ++e) t[e] = INF;
    auto insert = [&](float x) {
#pragma unroll
        for (int e = KW - 1; e >= 1; --e) t[e] = __builtin_amdgcn_fmed3f(x, t[e - 1], t[e]);
        t[0] = __builtin_fminf(t[0], x);
    };
    for (int i0 = wave; i0 < P.M; i0 += 32) {             // eight of the wave's rows in flight
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = i0 + 4 * e;
            x[e] = (ok && i < P.M) ? C[(size_t)i * P.pitchC] : INF;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) insert(x[e]);
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < KW; ++e) part[wave - 1][e][lane] = t[e];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int e = 0; e < KW; ++e) insert(part[w][e][lane]);
    const int kk = kw < P.M ? kw : P.M;
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < KW; ++e) sum += e < kk ? t[e] : 0.0f;        // ascending
    if (ok) stat[P.offS + s * ef_s_stride(P) + 2 * (int64_t)P.pitchT + j] = sum / (float)kk;
}

// ------------------------------------------------------------------------------------
// E3: fused = exp(-(W0 + W1 + W2)), elementwise over the pair's M x N cells
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ef_fuse_kernel(const EfPair *__restrict__ pd, float *__restrict__ scratch,
                                                      const float *__restrict__ stat, int exact)
{
    const EfPair P = pd[blockIdx.y];
    const int i = blockIdx.x;
    if (i >= P.M) return;
    float *F = scratch + ef_f_off(P) + (size_t)i * P.pitchC;
    float r[3];
    const float *c[3];
    const float *C[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float *S = stat + P.offS + s * ef_s_stride(P);
        r[s] = S[P.pitchT + i];
        c[s] = S + 2 * P.pitchT;
        C[s] = scratch + ef_c_off(P, s) + (size_t)i * P.pitchC;
    }
    for (int j = threadIdx.x; j < P.N; j += 256) {
        float wsum = 0.0f;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            wsum += exact ? ef_wcsm_weight_exact(r[s], c[s][j], C[s][j]) : ef_wcsm_weight(r[s], c[s][j], C[s][j]);
        }
        F[j] = exact ? ef_fused_value_exact(wsum) : ef_fused_value(wsum);
    }
}

// ------------------------------------------------------------------------------------
// E4: constrained Smith-Waterman, exact in integer tenths.  With T[i][j] = S[i+1][j+1] of the
// reference and U = T + delta(B) (delta = 0 if B else -7):
//   T[i][j] = max(0, mv(B[i][j]) + max(U[i-1][j-1], U[i-2][j-1], U[i-1][j-2])),  i, j >= 2,
//   i <= M-2, j <= N-2;  T = 0 (and U = delta(B)) in rows / columns 0, 1;  score = max T / 10.
// One wave per matrix; lane owns CPL contiguous columns (CPL = 8: N <= 512, CPL = 16: N <= 1024).  B_ij from (t_i, jcut_i), see ef_s_stride.
// src: 0..2 = feature CSM, 3 = fused matrix.  out[pair * 4 + src].
// ------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(64) void sw_kernel(const EfPair *__restrict__ pd, const float *__restrict__ scratch,
                                                const float *__restrict__ stat, float *__restrict__ out, int src_base)
{
    const int lane = threadIdx.x;
    const EfPair P = pd[blockIdx.x];
    const int src = src_base + blockIdx.y;
    const int M = P.M, N = P.N, pitch = P.pitchC;
    const float *C = scratch + (src < 3 ? ef_c_off(P, src) : ef_f_off(P));
    const float *thr = stat + P.offS + src * ef_s_stride(P);
    const int *jcut = reinterpret_cast<const int *>(thr) + ef_jcut_off(P, src);
    float result = 0.0f;
    if (M >= 4 && N >= 4) {
        int U1[CPL], U2[CPL];      // U of rows i-1, i-2
        const int prev = (lane + 63) & 63;
        const int j0 = CPL * lane;
        // rows 0 and 1: T = 0, U = delta(B)
        auto load_b = [&](int row, bool (&b)[CPL]) {
            const float t = thr[row];
            const int jc = jcut[row];
            float d[CPL];
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j0 + 4 * q < pitch) v = *reinterpret_cast<const float4 *>(C + (size_t)row * pitch + j0 + 4 * q);
                d[4 * q + 0] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int e = 0; e < CPL; ++e) b[e] = (j0 + e < N) && (d[e] < t || (d[e] == t && j0 + e <= jc));
        };
        bool b[CPL];
        load_b(0, b);
#pragma unroll
        for (int e = 0; e < CPL; ++e) U2[e] = b[e] ? 0 : -7;
        load_b(1, b);
#pragma unroll
        for (int e = 0; e < CPL; ++e) U1[e] = b[e] ? 0 : -7;
        int best = 0;
        // The rows of the matrix arrive through a register ring, SW_PF rows ahead of the row the recursion works
        // on: one wave walks one matrix, every row is a dependent trip to L2 / HBM otherwise (measured: 0.95 ms
        // per 1984 matrices of 400 x 400, ~ 2 us per row, with only two waves per SIMD to hide it).
        constexpr int SW_PF = 8;
        float4 ring[SW_PF][CPL / 4];
        float tring[SW_PF];
        int jring[SW_PF];
        auto issue = [&](int row, float4 (&dst)[CPL / 4], float &t, int &jc) {
            const int r = row < M ? row : M - 1;            // (rows past the last one: a valid address, never used)
            t = thr[r];
            jc = jcut[r];
#pragma unroll
            for (int q = 0; q < CPL / 4; ++q) {
                dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j0 + 4 * q < pitch) dst[q] = *reinterpret_cast<const float4 *>(C + (size_t)r * pitch + j0 + 4 * q);
            }
        };
#pragma unroll
        for (int sl = 0; sl < SW_PF; ++sl) issue(2 + sl, ring[sl], tring[sl], jring[sl]);
        for (int i0 = 2; i0 <= M - 2; i0 += SW_PF) {
#pragma unroll
            for (int sl = 0; sl < SW_PF; ++sl) {
                const int i = i0 + sl;
                if (i <= M - 2) {                            // wave-uniform
                    {
                        const float t = tring[sl];
                        const int jc = jring[sl];
#pragma unroll
                        for (int q = 0; q < CPL / 4; ++q) {
                            const float4 v = ring[sl][q];
                            const float d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) {
                                const int e = 4 * q + e4;
                                b[e] = (j0 + e < N) && (d[e4] < t || (d[e4] == t && j0 + e <= jc));
                            }
                        }
                    }
                    issue(i + SW_PF, ring[sl], tring[sl], jring[sl]);
                    const int l1a = lane_prev_i(U1[CPL - 1]), l1b = lane_prev_i(U1[CPL - 2]), l2a = lane_prev_i(U2[CPL - 1]);
                    int Tn[CPL];
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        const int c2 = (e >= 1) ? U1[e - 1] : l1a;                        // U[i-1][j-1]
                        const int c3 = (e >= 1) ? U2[e - 1] : l2a;                        // U[i-2][j-1]
                        const int c4 = (e >= 2) ? U1[e - 2] : (e == 1 ? l1a : l1b);       // U[i-1][j-2]
                        int mx = c2 > c3 ? c2 : c3;
                        mx = mx > c4 ? mx : c4;
                        int t = (b[e] ? 10 : -10) + mx;
                        t = t > 0 ? t : 0;
                        const int j = j0 + e;
                        if (j < 2) t = 0;                  // columns 0, 1 (lane 0 only; its shuffled inputs are unused)
                        Tn[e] = t;
                        if (j <= N - 2) best = best > t ? best : t;
                    }
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        U2[e] = U1[e];
                        U1[e] = Tn[e] + (b[e] ? 0 : -7);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int t = __shfl_xor(best, o, 64);
            best = best > t ? best : t;
        }
        result = (float)best / 10.0f;
    }
    if (lane == 0) out[(size_t)blockIdx.x * 4 + src] = result;
}

// ------------------------------------------------------------------------------------
// E4b: the same recursion on the BINARISED rows the selection kernels leave behind (round 3): a lane's CPL columns of a
// row are CPL bits -- one byte (CPL = 8) or two (16) -- instead of CPL floats, a threshold and a tie column; the four
// matrices of a pair in one launch.  out[pair * 4 + src].
// ------------------------------------------------------------------------------------
template <int CPL>
#ifndef ACX_SW_BITS_WAVES
#define ACX_SW_BITS_WAVES 1
#endif
__global__ __launch_bounds__(64, ACX_SW_BITS_WAVES) void sw_bits_kernel(const EfPair *__restrict__ pd, const unsigned *__restrict__ bits,
                                                     float *__restrict__ out, int src_base)
{
    const int lane = threadIdx.x;
    const EfPair P = pd[blockIdx.x];
    const int src = src_base + blockIdx.y;
    const int M = P.M, N = P.N, pitch = P.pitchC;
    float result = 0.0f;
    if (M >= 4 && N >= 4) {
        typedef unsigned short bits_t;                               // (CPL = 8 uses the low byte)
        const unsigned char *rows = reinterpret_cast<const unsigned char *>(bits + P.offB + (int64_t)src * M * (pitch >> 5));
        const int rowbytes = pitch >> 3;
        const int j0 = CPL * lane;
        const bool inrow = j0 < pitch;
        auto load = [&](int row) -> unsigned {
            const int r = row < M ? row : M - 1;                     // (rows past the last one: a valid address, never used)
            if (!inrow) return 0u;
            if (CPL == 8) return rows[(size_t)r * rowbytes + lane];
            return *reinterpret_cast<const bits_t *>(rows + (size_t)r * rowbytes + 2 * lane);
        };
        int U1[CPL], U2[CPL];      // U of rows i-1, i-2
        const int prev = (lane + 63) & 63;
        unsigned w = load(0);
#pragma unroll
        for (int e = 0; e < CPL; ++e) U2[e] = ((w >> e) & 1u) ? 0 : -7;
        w = load(1);
#pragma unroll
        for (int e = 0; e < CPL; ++e) U1[e] = ((w >> e) & 1u) ? 0 : -7;
        int best = 0;
        constexpr int SW_PF = 8;
        unsigned ring[SW_PF];
#pragma unroll
        for (int sl = 0; sl < SW_PF; ++sl) ring[sl] = load(2 + sl);
        for (int i0 = 2; i0 <= M - 2; i0 += SW_PF) {
#pragma unroll
            for (int sl = 0; sl < SW_PF; ++sl) {
                const int i = i0 + sl;
                if (i <= M - 2) {                            // wave-uniform
                    const unsigned wb = ring[sl];
                    ring[sl] = load(i + SW_PF);
                    bool b[CPL];
#pragma unroll
                    for (int e = 0; e < CPL; ++e) b[e] = ((wb >> e) & 1u) != 0u;
                    const int l1a = lane_prev_i(U1[CPL - 1]), l1b = lane_prev_i(U1[CPL - 2]), l2a = lane_prev_i(U2[CPL - 1]);
                    int Tn[CPL];
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        const int c2 = (e >= 1) ? U1[e - 1] : l1a;                        // U[i-1][j-1]
                        const int c3 = (e >= 1) ? U2[e - 1] : l2a;                        // U[i-2][j-1]
                        const int c4 = (e >= 2) ? U1[e - 2] : (e == 1 ? l1a : l1b);       // U[i-1][j-2]
                        int mx = c2 > c3 ? c2 : c3;
                        mx = mx > c4 ? mx : c4;
                        int t = (b[e] ? 10 : -10) + mx;
                        t = t > 0 ? t : 0;
                        const int j = j0 + e;
                        if (j < 2) t = 0;                  // columns 0, 1 (lane 0 only; its shuffled inputs are unused)
                        Tn[e] = t;
                        if (j <= N - 2) best = best > t ? best : t;
                    }
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        U2[e] = U1[e];
                        U1[e] = Tn[e] + (b[e] ? 0 : -7);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int t = __shfl_xor(best, o, 64);
            best = best > t ? best : t;
        }
        result = (float)best / 10.0f;
    }
    if (lane == 0) out[(size_t)blockIdx.x * 4 + src] = result;
}

// ------------------------------------------------------------------------------------
// E4c: E4b in PACKED 16-bit integers (round 4; the pattern of qmax_bits_h16_kernel): scores are tenths, at most 10 min(M, N)
// <= 10 240 for rows of <= 1024 cells, and U >= -7, so V = U + 7 fits the unsigned half of a register and one v_pk_* instruction
// updates two cells.  Register k of a lane holds its columns k (low half) and k + CPL / 2 (high half): the (i-1, j-1), (i-2, j-1),
// (i-1, j-2) predecessors of register k are registers k - 1 / k - 2 of the two previous rows (the first two stitched from the
// left neighbour lane by DPP + one funnel shift).  With the bias, T = max(max(U) +- 10, 0) = sat_sub(max(V) + 20 bit, 17) (an
// unsigned saturating subtract) and the new V = T + 7 bit: 11 packed instructions per two cells.  The same integers as E4b.
// ------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(64) void sw_bits_h16_kernel(const EfPair *__restrict__ pd, const unsigned *__restrict__ bits,
                                                         float *__restrict__ out, int src_base)
{
    constexpr int NR = CPL / 2;
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x;
    const EfPair P = pd[blockIdx.x];
    const int src = src_base + blockIdx.y;
    const int M = P.M, N = P.N, pitch = P.pitchC;
    float result = 0.0f;
    if (M >= 4 && N >= 4) {
        const unsigned char *rows = reinterpret_cast<const unsigned char *>(bits + P.offB + (int64_t)src * M * (pitch >> 5));
        const int rowbytes = pitch >> 3;
        const int j0 = CPL * lane;
        const bool inrow = j0 < pitch;
        auto load = [&](int row) -> unsigned {
            const int r = row < M ? row : M - 1;                     // (rows past the last one: a valid address, never used)
            if (!inrow) return 0u;
            if (CPL == 8) return rows[(size_t)r * rowbytes + lane];
            return *reinterpret_cast<const unsigned short *>(rows + (size_t)r * rowbytes + 2 * lane);
        };
        // a lane's CPL bits with the upper half moved to bit 16: the bits of columns k and k + NR are bit 0 of the halves of (w >> k)
        auto arrange = [&](unsigned v) -> unsigned { return (v & ((1u << NR) - 1u)) | (((v >> NR) & ((1u << NR) - 1u)) << 16); };
        auto spread = [&](unsigned va, int k) -> unsigned { return (va >> k) & 0x00010001u; };
        // columns that may hold a score (j >= 2) / that count for the maximum (j <= N - 2), as half masks per register
        unsigned cm[NR], bm[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int ja = j0 + k, jb = j0 + k + NR;
            cm[k] = (ja >= 2 ? 0xffffu : 0u) | (jb >= 2 ? 0xffff0000u : 0u);
            bm[k] = ((ja >= 2 && ja <= N - 2) ? 0xffffu : 0u) | ((jb >= 2 && jb <= N - 2) ? 0xffff0000u : 0u);
        }
        const u16x2 seven = {7, 7}, twenty = {20, 20}, seventeen = {17, 17};
        unsigned V1[NR], V2[NR];                                       // V = U + 7 of rows i-1, i-2
        {
            const unsigned w0 = arrange(load(0)), w1 = arrange(load(1));
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                V2[k] = __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2, spread(w0, k)) * seven);
                V1[k] = __builtin_bit_cast(unsigned, __builtin_bit_cast(u16x2, spread(w1, k)) * seven);
            }
        }
        i16x2 best = {0, 0};
        // one row: VA = row i-1, VB = row i-2 (overwritten with row i)
        auto dp_row = [&](unsigned wb, unsigned (&VA)[NR], unsigned (&VB)[NR]) {
            const unsigned w = arrange(wb);
            // the left neighbour's columns CPL-1 / CPL-2 are the HIGH halves of its registers NR-1 / NR-2 (lane 0: zeros, masked)
            const unsigned nA1 = lane_prev_u(VA[NR - 1]), nA2 = lane_prev_u(VA[NR - 2]), nB1 = lane_prev_u(VB[NR - 1]);
            // register "-1": low half = column -1 (the neighbour's), high half = column NR - 1 (own low half of register NR - 1)
            const unsigned a_m1 = __builtin_amdgcn_alignbit(VA[NR - 1], nA1, 16);
            const unsigned a_m2 = __builtin_amdgcn_alignbit(VA[NR - 2], nA2, 16);
            const unsigned b_m1 = __builtin_amdgcn_alignbit(VB[NR - 1], nB1, 16);
#pragma unroll
            for (int k = NR - 1; k >= 0; --k) {
                const u16x2 c2 = __builtin_bit_cast(u16x2, k >= 1 ? VA[k - 1] : a_m1);                          // (i-1, j-1)
                const u16x2 c3 = __builtin_bit_cast(u16x2, k >= 1 ? VB[k - 1] : b_m1);                          // (i-2, j-1)
                const u16x2 c4 = __builtin_bit_cast(u16x2, k >= 2 ? VA[k - 2] : (k == 1 ? a_m1 : a_m2));        // (i-1, j-2)
                const u16x2 mx = __builtin_elementwise_max(__builtin_elementwise_max(c2, c3), c4);
                const u16x2 sp = __builtin_bit_cast(u16x2, spread(w, k));
                // T = max(max(U) + (bit ? 10 : -10), 0) with U = V - 7
                const u16x2 tt = __builtin_elementwise_sub_sat(sp * twenty + mx, seventeen);
                const unsigned t = __builtin_bit_cast(unsigned, tt) & cm[k];                                     // columns 0, 1 stay 0
                best = __builtin_elementwise_max(best, __builtin_bit_cast(i16x2, t & bm[k]));
                VB[k] = __builtin_bit_cast(unsigned, sp * seven + __builtin_bit_cast(u16x2, t));                 // V = T + (bit ? 7 : 0)
            }
        };
        constexpr int SW_PF = 8;
        unsigned ring[SW_PF];
#pragma unroll
        for (int sl = 0; sl < SW_PF; ++sl) ring[sl] = load(2 + sl);
        for (int i0 = 2; i0 <= M - 2; i0 += SW_PF) {
#pragma unroll
            for (int sl = 0; sl < SW_PF; ++sl) {
                const int i = i0 + sl;
                if (i <= M - 2) {                            // wave-uniform
                    const unsigned wb = ring[sl];
                    ring[sl] = load(i + SW_PF);
                    if (sl & 1) dp_row(wb, V2, V1); else dp_row(wb, V1, V2);      // (i0 is even and SW_PF is even: row i-1 is V1 for even sl)
                }
            }
        }
        int bh = best.x > best.y ? best.x : best.y;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int t = __shfl_xor(bh, o, 64);
            bh = bh > t ? bh : t;
        }
        result = (float)bh / 10.0f;
    }
    if (lane == 0) out[(size_t)blockIdx.x * 4 + src] = result;
}

// ------------------------------------------------------------------------------------
// E2 / E4 for tracks of any length (more than 1024 blocks: a row no longer fits a wave's registers).
// Same results as the register-resident kernels, rows streamed from HBM / L2.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ef_key(float v)
{
    const unsigned u = __float_as_uint(v);
    return (u >> 31) ? ~u : (u | 0x80000000u);         // monotone: smaller float -> smaller key
}
__device__ __forceinline__ float ef_unkey(unsigned k)
{
    return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k);
}
// k-th smallest (0-based) of a row of n floats: binary search on the key, ballots count
__device__ __forceinline__ float ef_select_stream(const float *__restrict__ v, int n, int k, int lane)
{
    unsigned lo = 0u, hi = 0xffffffffu;
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int tot = 0;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            tot += __popcll(__ballot(j < n && ef_key(v[j]) <= mid));
        }
        if (tot >= k + 1) hi = mid; else lo = mid + 1;
    }
    return ef_unkey(lo);
}

__global__ __launch_bounds__(256) void ef_rowstat_long_kernel(const EfPair *__restrict__ pd, const float *__restrict__ scratch,
                                                              float *__restrict__ stat, int mode, int kw)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const EfPair P = pd[blockIdx.y];
    const int s = blockIdx.z;
    const int nrows = mode == 1 ? P.N : P.M;
    const int n = mode == 1 ? P.M : P.N;
    const int pitch = mode == 1 ? P.pitchT : P.pitchC;
    const int row = blockIdx.x * 4 + wave;
    if (row >= nrows) return;
    const int64_t base = mode == 0 ? ef_c_off(P, s) : (mode == 1 ? ef_ct_off(P, s) : ef_f_off(P));
    const float *v = scratch + base + (size_t)row * pitch;
    const float INF = __builtin_inff();
    float *S = stat + P.offS + (mode == 2 ? 3 * ef_s_stride(P) : s * ef_s_stride(P));
    if (mode != 1) {
        const int kb = P.kbin;
        float t;
        int jcut = 0x7fffffff;
        if (kb <= 0) t = -INF;
        else if (kb >= n) t = INF;
        else {
            t = ef_select_stream(v, n, kb - 1, lane);
            int lt = 0, eq = 0;
            for (int j0 = 0; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                const float x = j < n ? v[j] : INF;
                lt += __popcll(__ballot(j < n && x < t));
                eq += __popcll(__ballot(j < n && x == t));
            }
            const int budget = kb - lt;
            if (eq > budget) {                       // ties at the k-th value: taken in column order
                int seen = 0;
                for (int j0 = 0; j0 < n; j0 += 64) {
                    const int j = j0 + lane;
                    const unsigned long long m = __ballot(j < n && v[j] == t);
                    const int c = __popcll(m);
                    if (seen + c >= budget) {
                        // the (budget - seen)-th set bit of m
                        unsigned long long mm = m;
                        for (int q = 1; q < budget - seen; ++q) mm &= mm - 1;
                        jcut = j0 + (__ffsll((long long)mm) - 1);
                        break;
                    }
                    seen += c;
                }
            }
        }
        if (lane == 0) {
            S[row] = t;
            reinterpret_cast<int *>(S)[ef_jcut_off(P, mode == 2 ? 3 : s) + row] = jcut;
        }
    }
    if (mode != 2) {
        const int kk = kw < n ? kw : n;
        const float vk = ef_select_stream(v, n, kk - 1, lane);
        float acc = 0.0f;
        int cnt = 0;
        for (int j = lane; j < n; j += 64) {
            const float x = v[j];
            if (x < vk) { acc += x; ++cnt; }
        }
        const int tot = wave_sum_i(cnt);
        float sm = acc;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sm += __shfl_xor(sm, o, 64);
        const float m = (sm + (float)(kk - tot) * vk) / (float)kk;
        if (lane == 0) S[(mode == 0 ? P.pitchT : 2 * P.pitchT) + row] = m;
    }
}

// Constrained Smith-Waterman in strips of 64 x 16 = 1024 columns; for every row a strip leaves
// U[i][c - 1], U[i][c - 2] (c = first column of the next strip) in the pair's record area, which
// lane 0 of the next strip reads in place of the matrix edge.  Records: rec[(strip & 1)][row] (int2).
__global__ __launch_bounds__(64) void sw_long_kernel(const EfPair *__restrict__ pd, const float *__restrict__ scratch,
                                                     float *__restrict__ stat, float *__restrict__ out, int src_base)
{
    constexpr int CPL = 16;
    const int lane = threadIdx.x;
    const EfPair P = pd[blockIdx.x];
    const int src = src_base + blockIdx.y;
    const int M = P.M, N = P.N, pitch = P.pitchC;
    const float *C = scratch + (src < 3 ? ef_c_off(P, src) : ef_f_off(P));
    const float *thr = stat + P.offS + src * ef_s_stride(P);
    const int *jcut = reinterpret_cast<const int *>(thr) + ef_jcut_off(P, src);
    int2 *rec = reinterpret_cast<int2 *>(stat + P.offS + ef_rec_off(P, src));
    float result = 0.0f;
    if (M >= 4 && N >= 4) {
        const int nstrips = (N + 64 * CPL - 1) / (64 * CPL);
        const int prev = (lane + 63) & 63;
        int best = 0;
        for (int st = 0; st < nstrips; ++st) {
            const int j0 = st * 64 * CPL + CPL * lane;
            const int2 *rin = rec + (size_t)((st + 1) & 1) * P.pitchT;
            int2 *rout = rec + (size_t)(st & 1) * P.pitchT;
            const bool more = st + 1 < nstrips;
            int U1[CPL], U2[CPL];
            auto load_b = [&](int row, bool (&b)[CPL]) {
                const float t = thr[row];
                const int jc = jcut[row];
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    const int j = j0 + e;
                    const float d = j < N ? C[(size_t)row * pitch + j] : 0.0f;
                    b[e] = (j < N) && (d < t || (d == t && j <= jc));
                }
            };
            bool b[CPL];
            load_b(0, b);
#pragma unroll
            for (int e = 0; e < CPL; ++e) U2[e] = b[e] ? 0 : -7;
            if (more && lane == 63) rout[0] = make_int2(U2[CPL - 1], U2[CPL - 2]);
            load_b(1, b);
#pragma unroll
            for (int e = 0; e < CPL; ++e) U1[e] = b[e] ? 0 : -7;
            if (more && lane == 63) rout[1] = make_int2(U1[CPL - 1], U1[CPL - 2]);
            int2 recB = make_int2(0, 0);                        // record of row i - 2
            if (st > 0) recB = rin[0];
            for (int i = 2; i <= M - 2; ++i) {
                load_b(i, b);
                int l1a = lane_prev_i(U1[CPL - 1]), l1b = lane_prev_i(U1[CPL - 2]), l2a = lane_prev_i(U2[CPL - 1]);
                int2 recA = make_int2(0, 0);                    // record of row i - 1
                if (st > 0) recA = rin[i - 1];
                if (lane == 0) { l1a = recA.x; l1b = recA.y; l2a = recB.x; }
                recB = recA;
                int Tn[CPL];
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    const int c2 = (e >= 1) ? U1[e - 1] : l1a;
                    const int c3 = (e >= 1) ? U2[e - 1] : l2a;
                    const int c4 = (e >= 2) ? U1[e - 2] : (e == 1 ? l1a : l1b);
                    int mx = c2 > c3 ? c2 : c3;
                    mx = mx > c4 ? mx : c4;
                    int t = (b[e] ? 10 : -10) + mx;
                    t = t > 0 ? t : 0;
                    const int j = j0 + e;
                    if (j < 2) t = 0;
                    Tn[e] = t;
                    if (j <= N - 2) best = best > t ? best : t;
                }
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    U2[e] = U1[e];
                    U1[e] = Tn[e] + (b[e] ? 0 : -7);
                }
                if (more && lane == 63) rout[i] = make_int2(U1[CPL - 1], U1[CPL - 2]);
            }
            __threadfence();                                    // the next strip reads this one's records
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int t = __shfl_xor(best, o, 64);
            best = best > t ? best : t;
        }
        result = (float)best / 10.0f;
    }
    if (lane == 0) out[(size_t)blockIdx.x * 4 + src] = result;
}

}  // namespace acx
