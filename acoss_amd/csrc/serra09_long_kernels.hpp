// Serra09 kernels for LONG tracks (gfx950): pairs whose embedded matrix has rows of more than
// 2041 cells -- more than the band kernel of serra09_kernels.hpp can hold in one wave's registers --
// or a delay-embedding stack of more than 16 frames.  Same arithmetic spec, same bits (DESIGN.md
// section 2; oracle/acx_oracle.c), any matrix size; the reference accepts any track length
// (acoss/algorithms/rqa_serra09.py:55-69 hands whatever load_features returned to essentia).
//
//   L1 csm_long_kernel       64 x 64 tiles of squared embedded distances: frame Gram on the matrix
//                            cores, doubling-tree window sums for a run-time stack size m <= 33;
//                            writes D2 (rows = query frames) and its transpose to HBM
//   L2 rowsel_long_kernel    exact kappa-percentile threshold of every row of D2 and of D2^T: one
//                            wave per row, the row is STREAMED from HBM / L2 once per narrowing pass
//   L3 binarise_long_kernel  R = [d2 <= thr_row][d2 <= thr_col] -> the recurrence bitmap in the band
//                            pipeline's layout
//   L4 qmax_bits_long_kernel Qmax / Dmax over the bitmap in strips of 2048 columns; the two rightmost
//                            columns of a strip travel to the next one through HBM
// These pairs are rare (a track of more than 2050 pooled frames is 16 minutes of audio at the
// default profile), so the kernels are written for generality, not for the roofline.
#pragma once
#include "serra09_kernels.hpp"

namespace acx {

constexpr int LT = 64;                 // cell tile edge
constexpr int LHALO = 32;              // frames of halo: m - 1 <= 32
constexpr int LST = LT + LHALO;        // frames per side of a tile's Gram
constexpr int LSP = LST + 1;           // pitch of the Gram tile in LDS
constexpr int MAX_M_LONG = LHALO + 1;  // largest stack size

// balanced tree over W values p[0], p[stride], ... (W a power of two) == tree_w<W> on a gathered array
template <int W>
__device__ __forceinline__ float tree_w_strided(const float *p, int stride)
{
    if constexpr (W == 1) {
        return p[0];
    } else {
        const float a = tree_w_strided<W / 2>(p, stride);
        const float b = tree_w_strided<W / 2>(p + (W / 2) * stride, stride);
        return a + b;
    }
}
__device__ __forceinline__ float tree_block_rt(const float *p, int stride, int W)
{
    switch (W) {
    case 1: return p[0];
    case 2: return tree_w_strided<2>(p, stride);
    case 4: return tree_w_strided<4>(p, stride);
    case 8: return tree_w_strided<8>(p, stride);
    case 16: return tree_w_strided<16>(p, stride);
    default: return tree_w_strided<32>(p, stride);
    }
}
// tree_sum<m> (serra09_kernels.hpp) for a run-time m: the block of the highest set bit of m first,
// then the lower set bits high to low, each a balanced tree over its own consecutive values
__device__ __forceinline__ float tree_sum_rt(const float *p, int stride, int m)
{
    int B = 1 << (31 - __clz(m));
    float acc = tree_block_rt(p, stride, B);
    int off = B;
    for (B >>= 1; B >= 1; B >>= 1) {
        if (m & B) {
            const float t = tree_block_rt(p + off * stride, stride, B);
            acc = acc + t;
            off += B;
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------
// L1: squared embedded distances, one 64 x 64 tile per workgroup (4 waves).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csm_long_kernel(const float *__restrict__ pool,
                                                       const int64_t *__restrict__ toff,
                                                       const PairDesc *__restrict__ pd,
                                                       float *__restrict__ scratch,
                                                       int tiles_x, int oti_target, int m)
{
    __shared__ float Qs[NBIN][LST];     // chroma tiles, bin-major (conflict-free MFMA operand reads)
    __shared__ float Rs[NBIN][LST];
    __shared__ float NQ[LST], NR[LST];
    __shared__ float XX[LT], YY[LT];
    __shared__ float S[LST * LSP];      // frame Gram tile; later the transpose stage

    const PairDesc P = pd[blockIdx.y];
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int i0 = ty * LT, j0 = tx * LT;
    if (i0 >= P.Mq || j0 >= P.Mr) return;   // block-uniform

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *qf = pool + toff[P.q] * NBIN;
    const float *rf = pool + toff[P.r] * NBIN;
    const int rotq = (oti_target == 1) ? P.oti : 0;
    const int rotr = (oti_target == 0) ? P.oti : 0;

    // chroma, rotated (roll right: dst bin (c + s) % 12 <- src bin c)
    for (int idx = tid; idx < LST * NBIN; idx += 256) {
        const int a = idx / NBIN, c = idx - a * NBIN;
        const int fq = i0 + a, fr = j0 + a;
        const float vq = (fq < P.Tq) ? qf[(size_t)fq * NBIN + c] : 0.0f;
        const float vr = (fr < P.Tr) ? rf[(size_t)fr * NBIN + c] : 0.0f;
        int cq = c + rotq; if (cq >= NBIN) cq -= NBIN;
        int cr = c + rotr; if (cr >= NBIN) cr -= NBIN;
        Qs[cq][a] = vq;
        Rs[cr][a] = vr;
    }
    __syncthreads();

    // frame norms: fmaf chain over the bins in rotated order
    if (tid < 2 * LST) {
        const bool isq = tid < LST;
        const int a = isq ? tid : tid - LST;
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < NBIN; ++c) {
            const float v = isq ? Qs[c][a] : Rs[c][a];
            acc = fmaf(v, v, acc);
        }
        if (isq) NQ[a] = acc; else NR[a] = acc;
    }

    // frame Gram on the matrix cores: 6 x 6 blocks of 16 x 16, K = 12 in 3 k-steps (== the fmaf chain)
    {
        const int lr = lane & 15, lk = lane >> 4;
        constexpr int NB = LST / 16;
        for (int t = wave; t < NB * NB; t += 4) {
            const int ta = t / NB, tb = t - ta * NB;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                const float av = Qs[4 * kb + lk][16 * ta + lr];
                const float bv = Rs[4 * kb + lk][16 * tb + lr];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                S[(16 * ta + 4 * lk + reg) * LSP + 16 * tb + lr] = acc[reg];
        }
    }
    __syncthreads();

    // embedded norms of the tile's rows / columns
    if (tid < 2 * LT) {
        const bool isq = tid < LT;
        const int a = isq ? tid : tid - LT;
        const float v = tree_sum_rt((isq ? NQ : NR) + a, 1, m);
        if (isq) XX[a] = v; else YY[a] = v;
    }
    __syncthreads();

    // window sums down the diagonal + distance: thread = (column c, 16-row quarter rq)
    const int c = lane, rq = wave;
    float o[16];
    {
        const float yy = YY[c];
        const bool colok = (j0 + c) < P.Mr;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int a = 16 * rq + t;
            const float xy = tree_sum_rt(S + a * LSP + c, LSP + 1, m);
            const float t1 = 2.0f * xy;
            const float t2 = XX[a] - t1;
            float t3 = t2 + yy;
            if (!(t3 > 0.0f)) t3 = 0.0f;
            o[t] = colok ? t3 : __builtin_inff();
        }
    }
    {
        float *D = scratch + P.offD;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = i0 + 16 * rq + t;
            if (i < P.Mq) D[(size_t)i * P.pitchD + j0 + c] = o[t];
        }
    }
    __syncthreads();   // all Gram reads done -> S becomes the transpose stage O[64][65]
#pragma unroll
    for (int t = 0; t < 16; ++t) S[(16 * rq + t) * 65 + c] = o[t];
    __syncthreads();
    {
        float *DT = scratch + P.offL;
        const bool iok = (i0 + c) < P.Mq;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int b = 16 * rq + t;      // tile column -> row of D2^T
            const int j = j0 + b;
            const float v = S[c * 65 + b];
            if (j < P.Mr) DT[(size_t)j * P.pitchT + i0 + c] = iok ? v : __builtin_inff();
        }
    }
}

// ------------------------------------------------------------------------------------
// k-th smallest (0-based) of a row of n floats in HBM -- exact, any n.  Same narrowing scheme as
// wave_select_regs, but the row is streamed once per pass instead of living in registers:
// histogram the active value range into SEL_BINS linear bins, descend into the bin holding rank
// k until it holds <= 64 elements, rank those directly.  The bin of a value is a pure function of
// (value, range), so the histogram pass and the gather pass agree.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ SelectResult wave_select_stream(const float *__restrict__ v, int n, int k, unsigned *hist,
                                                           float *cand, unsigned *counter, int lane, bool want_next)
{
    using SG = SelGeom<SEL_BINS>;
    constexpr int BPL = SG::BPL;
    const float INF = __builtin_inff();
    float mn = INF, mx = -INF;
    for (int j = lane; j < n; j += 64) {
        const float x = v[j];
        mn = fminf(mn, x);
        mx = fmaxf(mx, x);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    int below = 0;
    float result = mn;
    for (int iter = 0; iter < 64; ++iter) {
        if (!(mn < mx)) { result = mn; break; }
        const float scale = (float)SEL_BINS / (mx - mn);
        auto bin_of = [&](float x) {
            int b = (int)((x - mn) * scale);
            return b > SEL_BINS - 1 ? SEL_BINS - 1 : b;
        };
        for (int b = lane * 4; b < SG::SLOTS; b += 256)
            *reinterpret_cast<uint4 *>(hist + b) = make_uint4(0, 0, 0, 0);
        if (lane == 0) *counter = 0u;
        wave_lds_fence();
        for (int j = lane; j < n; j += 64) {
            const float x = v[j];
            if (x >= mn && x <= mx)
                __hip_atomic_fetch_add(&hist[SG::slot(bin_of(x))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        wave_lds_fence();
        int hv[BPL];
        int lsum = 0;
#pragma unroll
        for (int e = 0; e < BPL; ++e) { hv[e] = (int)hist[(BPL + 1) * lane + e]; lsum += hv[e]; }
        const int incl = wave_incl_scan_i(lsum);
        const int target = k - below;
        const int L = __ffsll((long long)__ballot(incl > target)) - 1;
        const int excl = incl - lsum;
        int binsel_v = 0, cum_v = 0, cnt_v = 0;
        {
            int run = excl;
#pragma unroll
            for (int e = 0; e < BPL; ++e) {
                const bool here = cnt_v == 0 && run + hv[e] > target;
                binsel_v = here ? BPL * lane + e : binsel_v;
                cum_v = here ? run : cum_v;
                cnt_v = here ? hv[e] : cnt_v;
                run += hv[e];
            }
        }
        const int binsel = __builtin_amdgcn_readlane(binsel_v, L);
        const int cum = __builtin_amdgcn_readlane(cum_v, L);
        const int cnt = __builtin_amdgcn_readlane(cnt_v, L);
        if (cnt <= 64) {
            for (int j = lane; j < n; j += 64) {
                const float x = v[j];
                if (x >= mn && x <= mx && bin_of(x) == binsel) {
                    const unsigned pos = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    cand[pos & 63u] = x;
                }
            }
            wave_lds_fence();
            const float mine = (lane < cnt) ? cand[lane] : INF;
            int rank = 0;
#pragma unroll 1
            for (int t = 0; t < cnt; ++t) {
                const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), t));
                rank += (o < mine || (o == mine && t < lane)) ? 1 : 0;
            }
            const int want = target - cum;
            const int src = __ffsll((long long)__ballot(lane < cnt && rank == want)) - 1;
            result = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), src));
            wave_lds_fence();
            break;
        }
        float nmn = INF, nmx = -INF;
        for (int j = lane; j < n; j += 64) {
            const float x = v[j];
            if (x >= mn && x <= mx && bin_of(x) == binsel) { nmn = fminf(nmn, x); nmx = fmaxf(nmx, x); }
        }
        below += cum;
        mn = wave_min(nmn);
        mx = wave_max(nmx);
        result = mn;
        wave_lds_fence();
    }
    SelectResult res{result, 0, INF};
    if (want_next) {
        int cle = 0;
        float nx = INF;
        for (int j = lane; j < n; j += 64) {
            const float x = v[j];
            cle += (x <= result) ? 1 : 0;
            nx = fminf(nx, (x > result) ? x : INF);
        }
        res.cnt_le = wave_sum_i(cle);
        res.next = wave_min(nx);
    }
    return res;
}

// ------------------------------------------------------------------------------------
// L2: thresholds.  Row r < Mq: row r of D2 (a query frame against every reference frame) -> row
// threshold; row r >= Mq: row r - Mq of D2^T -> column threshold.  One wave per row.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowsel_long_kernel(const PairDesc *__restrict__ pd,
                                                          const float *__restrict__ scratch,
                                                          float *__restrict__ thr,
                                                          float kappa, int pct_mode, int inclusive)
{
    __shared__ __attribute__((aligned(16))) unsigned hist[4][SelGeom<SEL_BINS>::SLOTS];
    __shared__ float cand[4][64];
    __shared__ unsigned counter[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const PairDesc P = pd[blockIdx.y];
    const int r = blockIdx.x * 4 + wave;
    if (r >= P.Mq + P.Mr) return;   // wave-uniform; no workgroup barriers below
    const bool side = r >= P.Mq;
    const int row = side ? r - P.Mq : r;
    const int n = side ? P.Mq : P.Mr;
    const float *v = side ? scratch + P.offL + (size_t)row * P.pitchT : scratch + P.offD + (size_t)row * P.pitchD;

    // percentile position in f32 (oracle percentile_f32)
    const float kf = (n > 1) ? __fmul_rn((float)(n - 1), kappa) : __fmul_rn((float)n, kappa);
    const float fl = floorf(kf), ce = ceilf(kf);
    int ilo = (int)fl, ihi = (int)ce;
    ilo = ilo < 0 ? 0 : (ilo > n - 1 ? n - 1 : ilo);
    ihi = ihi < 0 ? 0 : (ihi > n - 1 ? n - 1 : ihi);
    int k = ilo;
    if (pct_mode == 3) {
        k = (int)floorf(__fadd_rn(kf, 0.5f));
        k = k > n - 1 ? n - 1 : k;
    }
    const bool interp = (pct_mode == 0 || pct_mode == 1);
    const SelectResult sr = wave_select_stream(v, n, k, hist[wave], cand[wave], &counter[wave], lane, interp);
    const float eps = percentile_eps(sr, pct_mode, ilo, ihi, kf, fl, ce);
    if (lane == 0) {
        float *X = thr + P.offX;
        const int o = side ? P.pitchT + row : row;
        X[o] = d2_threshold(eps, inclusive);
        X[P.pitchT + P.pitchD + o] = eps;
    }
}

// ------------------------------------------------------------------------------------
// L3: recurrence bitmap in the band pipeline's layout: word t of row i = columns
// [64 t - 7 + (i & 7), +64).  One wave per row.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void binarise_long_kernel(const PairDesc *__restrict__ pd,
                                                            const float *__restrict__ scratch,
                                                            const float *__restrict__ thr,
                                                            unsigned long long *__restrict__ bits)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const PairDesc P = pd[blockIdx.y];
    const int i = blockIdx.x * 4 + wave;
    if (i >= P.Mq) return;
    const float *X = thr + P.offX;
    const float thr_row = X[i];
    const float *tcol = X + P.pitchT;
    const float *D = scratch + P.offD + (size_t)i * P.pitchD;
    unsigned long long *rowbits = bits + P.offT + (size_t)i * P.nw;
    const int c0 = (i & (BAND - 1)) - (BAND - 1);
    for (int t = 0; t < P.nw; ++t) {
        const int col = 64 * t + c0 + lane;
        bool r = false;
        if (col >= 0 && col < P.Mr) r = D[col] <= fminf(thr_row, tcol[col]);
        const unsigned long long w = __ballot(r);
        if (lane == 0) rowbits[t] = w;
    }
}

// ------------------------------------------------------------------------------------
// L4: Qmax / Dmax on the recurrence bitmap in strips of 64 x 32 = 2048 columns.  Within a strip
// this is qmax_bits_kernel<.., 32>; for every row the strip leaves a record {Q[i][c-1], Q[i][c-2]
// and their penalised versions} of its two rightmost columns in HBM (c = first column of the next
// strip), which lane 0 of the next strip reads in place of the zeros of the matrix edge.
// Records: bnd[(strip & 1)][row] as float4, behind D2^T in the scratch arena.
// ------------------------------------------------------------------------------------
template <bool EQG, bool DMAX>
__global__ __launch_bounds__(64) void qmax_bits_long_kernel(const PairDesc *__restrict__ pd,
                                                            const unsigned long long *__restrict__ bits,
                                                            float *__restrict__ scratch,
                                                            float *__restrict__ out, int out_stride,
                                                            float go, float ge, int dp_start)
{
    constexpr int CPL = 32;
    const int lane = threadIdx.x;
    const PairDesc P = pd[blockIdx.x];
    int Me = P.Mq, Ne = P.Mr;
    if (dp_start == 3) { Me -= 1; Ne -= 1; }
    const int ndw = 2 * P.nw;
    const unsigned *rows = reinterpret_cast<const unsigned *>(bits + P.offT);
    float4 *bnd = reinterpret_cast<float4 *>(scratch + P.offL + (size_t)P.Mr * P.pitchT);
    const int nstrips = (Ne + 64 * CPL - 1) / (64 * CPL);
    const int prev = (lane + 63) & 63;
    float best = 0.0f;

    for (int s = 0; s < nstrips; ++s) {
        const int cbase = s * 64 * CPL;                              // first column of the strip
        const float4 *bin = bnd + (size_t)((s + 1) & 1) * P.Mq;       // records of strip s - 1
        float4 *bout = bnd + (size_t)(s & 1) * P.Mq;
        unsigned colmask = 0u;
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            const int j = cbase + CPL * lane + e;
            if (j >= 2 && j < Ne) colmask |= (1u << e);
        }
        float Q1[CPL], Q2[CPL];
        float P1[EQG ? 1 : CPL], P2[EQG ? 1 : CPL];
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            Q1[e] = 0.0f; Q2[e] = 0.0f;
            if constexpr (!EQG) { P1[e] = 0.0f; P2[e] = 0.0f; }
        }
        const int dw0 = (cbase + CPL * lane) >> 5;                   // (bit0 == 0: CPL == 32)
        const bool has0 = dw0 < ndw, has1 = dw0 + 1 < ndw;
        auto load_row = [&](int i, unsigned &d0, unsigned &d1) {
            d0 = 0u; d1 = 0u;
            if (i < Me) {
                const unsigned *r = rows + (size_t)i * ndw;
                if (has0) d0 = r[dw0];
                if (has1) d1 = r[dw0 + 1];
            }
        };
        auto row_bits = [&](int i, unsigned d0, unsigned d1) {
            const int sh = (BAND - 1) - (i & (BAND - 1));
            return __builtin_amdgcn_alignbit(d1, d0, sh);
        };
        auto left_bit = [&](int i) {                                 // R[i][cbase - 1], raw
            const int bp = cbase - 1 + (BAND - 1) - (i & (BAND - 1));
            return (rows[(size_t)i * ndw + (bp >> 5)] >> (bp & 31)) & 1u;
        };
        unsigned wprev = 0u;
        if constexpr (DMAX) {
            unsigned p0, p1;
            load_row(1, p0, p1);
            wprev = row_bits(1, p0, p1);
            if constexpr (!EQG) {
                load_row(0, p0, p1);
                const unsigned w0 = row_bits(0, p0, p1);
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    P1[e] = ((wprev >> e) & 1u) ? -go : -ge;
                    P2[e] = ((w0 >> e) & 1u) ? -go : -ge;
                }
            }
        }
        const bool more = s + 1 < nstrips;
        if (more && lane == 63) {          // rows 0 and 1 of Q are zero; their penalised versions as initialised
            if (Me > 0) bout[0] = make_float4(0.0f, 0.0f, EQG ? 0.0f : P2[EQG ? 0 : CPL - 1], EQG ? 0.0f : P2[EQG ? 0 : CPL - 2]);
            if (Me > 1) bout[1] = make_float4(0.0f, 0.0f, EQG ? 0.0f : P1[EQG ? 0 : CPL - 1], EQG ? 0.0f : P1[EQG ? 0 : CPL - 2]);
        }
        float4 recB = make_float4(0.f, 0.f, 0.f, 0.f);                // record of row i - 2
        if (s > 0 && Me > 0) recB = bin[0];
        auto dp_row = [&](int i, unsigned d0, unsigned d1, float (&QA)[CPL], float (&QB)[CPL],
                          float (&PA)[EQG ? 1 : CPL], float (&PB)[EQG ? 1 : CPL]) {
            const unsigned wraw = row_bits(i, d0, d1);
            float l1a = wave_shfl(QA[CPL - 1], prev), l1b = wave_shfl(QA[CPL - 2], prev), l2a = wave_shfl(QB[CPL - 1], prev);
            float p1a = 0.f, p1b = 0.f, p2a = 0.f;
            if constexpr (!EQG) {
                p1a = wave_shfl(PA[CPL - 1], prev); p1b = wave_shfl(PA[CPL - 2], prev); p2a = wave_shfl(PB[CPL - 1], prev);
            }
            float4 recA = make_float4(0.f, 0.f, 0.f, 0.f);            // record of row i - 1
            if (s > 0) recA = bin[i - 1];
            if (lane == 0) {
                l1a = recA.x; l1b = recA.y; l2a = recB.x;
                p1a = recA.z; p1b = recA.w; p2a = recB.z;
            }
            recB = recA;
            unsigned wleft = 0u;
            if constexpr (DMAX) {
                unsigned carry = (unsigned)__shfl((int)((wraw >> (CPL - 1)) & 1u), prev, 64);
                if (lane == 0) carry = s > 0 ? left_bit(i) : 0u;
                wleft = (wraw << 1) | carry;
            }
            qmax_cells<EQG, DMAX, CPL>(wraw, colmask, wprev, wleft, l1a, l1b, l2a, p1a, p1b, p2a, QA, QB, PA, PB, go, ge, best);
            if (more && lane == 63)
                bout[i] = make_float4(QB[CPL - 1], QB[CPL - 2], EQG ? 0.0f : PB[EQG ? 0 : CPL - 1], EQG ? 0.0f : PB[EQG ? 0 : CPL - 2]);
            if constexpr (DMAX) wprev = wraw;
        };
        for (int i = 2; i < Me; i += 2) {
            unsigned a0, a1;
            load_row(i, a0, a1);
            dp_row(i, a0, a1, Q1, Q2, P1, P2);
            if (i + 1 < Me) {
                load_row(i + 1, a0, a1);
                dp_row(i + 1, a0, a1, Q2, Q1, P2, P1);
            }
        }
        // the next strip reads this one's records: same wave, but through memory
        __threadfence();
    }
    best = wave_max(best);
    if (lane == 0) out[(size_t)blockIdx.x * out_stride] = best;
}

}  // namespace acx
