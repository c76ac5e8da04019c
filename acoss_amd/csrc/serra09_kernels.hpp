// Serra09 device kernels for gfx950 (MI355X / CDNA4).  Wave = 64 lanes.
//
// Per-pair chain (reference call site acoss/algorithms/rqa_serra09.py:55-69, arithmetic
// spec in DESIGN.md / oracle/acx_oracle.c):
//
// Production pipeline (band pipeline):
//   K0  oti_kernel        12-bin optimal transposition index per pair
//   K0b normtab_kernel    embedded norms per (track, rotation, frame), once per (pool, m)
//       rotpool_kernel    rotated frame pool (MFMA operands in chain order), once per upload
//   K1' band_kernel       role 1 then role 0: frame Gram on the matrix cores
//                         (v_mfma_f32_16x16x4_f32, K = 12 = 3 k-steps, exact f32 == fmaf chain),
//                         diagonal doubling-tree window sums, exact kappa-percentile thresholds
//                         per row (one pivot-filtered histogram pass), recurrence bitmap; D2 never reaches HBM
//   K3b qmax_bits_kernel  Qmax / Dmax row sweep over the recurrence bitmap, one wave per pair
// Tracks whose embedded matrix has rows of more than 2041 cells (or a stack size m > 16) take the
// streaming kernels of serra09_long_kernels.hpp (materialised D2 + D2^T) and rejoin this pipeline at
// the recurrence bitmap.
//
// Everything is f32; the operation ORDER is part of the spec (bit-exact parity with the
// oracle), so this file is compiled with -ffp-contract=off and uses explicit fmaf only
// where the spec says fma.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace acx {

constexpr int NBIN = 12;       // chroma bins
constexpr int SEL_BINS = 2048; // histogram bins of the generic percentile selection
constexpr int MAX_M = 16;      // largest stack size of the band kernel (larger m: long-track kernels)
constexpr int NGUARD = 80;     // +inf entries behind every row of the embedded-norm table (normtab_kernel)

// Position of the kappa-percentile in a row of n cells (oracle percentile_f32): the same for every row of a
// pair, so the host works it out once per pair (both row lengths) and the kernels read it from the PairDesc
// with scalar loads.
struct PctPos { float kf, fl, ce; int32_t ilo, ihi, k; };
__host__ __device__ inline PctPos pct_position(int n, float kappa, int pct_mode)
{
    PctPos p;
    p.kf = (n > 1) ? (float)(n - 1) * kappa : (float)n * kappa;        // (one f32 multiplication: nothing to contract)
    p.fl = floorf(p.kf); p.ce = ceilf(p.kf);
    int ilo = (int)p.fl, ihi = (int)p.ce;
    ilo = ilo < 0 ? 0 : (ilo > n - 1 ? n - 1 : ilo);
    ihi = ihi < 0 ? 0 : (ihi > n - 1 ? n - 1 : ihi);
    int k = ilo;
    if (pct_mode == 3) {
        k = (int)floorf(p.kf + 0.5f);
        k = k > n - 1 ? n - 1 : k;
    }
    p.ilo = ilo; p.ihi = ihi; p.k = k;
    return p;
}

struct PairDesc {
    int32_t q, r;          // track indices (query, reference)
    int32_t Tq, Tr;        // pooled lengths
    int32_t Mq, Mr;        // embedded lengths (rows, cols of the matrix)
    int32_t oti;           // filled by K0
    int32_t pitchD;        // row pitch of D2  (floats, multiple of 64, >= Mr)
    int32_t pitchT;        // row pitch of D2T (floats, multiple of 64, >= Mq)
    int32_t nw;            // 64-bit words per row of the recurrence bitmap
    int64_t offD;          // float offset of D2 in the scratch arena (debug entry point and long tracks only)
    int64_t offT;          // u64-word offset of the pair's recurrence bitmap (Mq rows x nw words) in the bit arena
    int64_t offX;          // float offset into the threshold arena:
                           //   [thr rows: pitchT][thr cols: pitchD][eps rows: pitchT][eps cols: pitchD]
    int64_t offL;          // long tracks: float offset of D2^T (Mr rows x pitchT) in the scratch arena,
                           // followed by the DP's strip-boundary records (2 x 4 floats per row)
    PctPos pos_q, pos_r;   // percentile position in a row of Mq cells (column pass) / of Mr cells (row pass)
    int64_t fq, fr;        // filled by K0: first frame of the query / reference track in the active pool (toff[q], toff[r])
    int64_t nq, nr;        //               and their rows of the embedded-norm table (noff[q], noff[r]; 0 without a table):
                           //               the band kernel starts its operand loads one dependent memory round trip earlier
};

__device__ __forceinline__ float wave_shfl(float v, int src)
{
    return __shfl(v, src, 64);
}
// The value of lane - 1 (lane 0: zero) by DPP wave_shr:1 -- the left neighbour of the alignment recursions.  __shfl(v, lane - 1)
// is a ds_bpermute_b32: an LDS round trip in the dependency chain of every DP row.
__device__ __forceinline__ int lane_prev_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ unsigned lane_prev_u(unsigned v) { return (unsigned)lane_prev_i((int)v); }
__device__ __forceinline__ float lane_prev_f(float v) { return __int_as_float(lane_prev_i(__float_as_int(v))); }

// ------------------------------------------------------------------------------------
// doubling-tree window sum (DESIGN.md "arithmetic spec"; oracle tree_sum)
// ------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ float tree_w(const float *s)
{
    if constexpr (W == 1) {
        return s[0];
    } else {
        float a = tree_w<W / 2>(s);
        float b = tree_w<W / 2>(s + W / 2);
        return a + b;
    }
}
template <int M, int B, int OFF>
__device__ __forceinline__ float tree_low(const float *s, float acc)
{
    if constexpr (B == 0) {
        return acc;
    } else if constexpr ((M & B) != 0) {
        float t = tree_w<B>(s + OFF);
        return tree_low<M, B / 2, OFF + B>(s, acc + t);
    } else {
        return tree_low<M, B / 2, OFF>(s, acc);
    }
}
constexpr int high_bit(int m)
{
    int hb = 1;
    while (hb * 2 <= m) hb *= 2;
    return hb;
}
template <int M>
__device__ __forceinline__ float tree_sum(const float *s)
{
    constexpr int HB = high_bit(M);
    float acc = tree_w<HB>(s);
    return tree_low<M, HB / 2, HB>(s, acc);
}

// ------------------------------------------------------------------------------------
// K0: OTI.  argmax_s <ga, roll(gb, s)>, s = 0..12, first max wins; separate mul / add.
// ------------------------------------------------------------------------------------
static __global__ void oti_kernel(PairDesc *pd, int B, const float *__restrict__ gch, int oti_on, int oti_target,
                                  const int64_t *__restrict__ toff, const int64_t *__restrict__ noff)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
    pd[p].fq = toff[pd[p].q]; pd[p].fr = toff[pd[p].r];
    pd[p].nq = noff ? noff[pd[p].q] : 0; pd[p].nr = noff ? noff[pd[p].r] : 0;
    int best = 0;
    if (oti_on) {
        const float *ga = gch + (size_t)NBIN * (oti_target == 0 ? pd[p].q : pd[p].r);
        const float *gb = gch + (size_t)NBIN * (oti_target == 0 ? pd[p].r : pd[p].q);
        float a[NBIN], b[NBIN];
        for (int c = 0; c < NBIN; ++c) { a[c] = ga[c]; b[c] = gb[c]; }
        float bestv = 0.0f;
        for (int s = 0; s <= NBIN; ++s) {
            float acc = 0.0f;
            for (int c = 0; c < NBIN; ++c) {
                float pr = __fmul_rn(a[c], b[(c - s + 2 * NBIN) % NBIN]);
                acc = __fadd_rn(acc, pr);
            }
            if (s == 0 || acc > bestv) { bestv = acc; best = s; }
        }
        best = best % NBIN;
    }
    pd[p].oti = best;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS operations of one wave execute in order; this keeps the compiler from reordering them
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Wave64 reductions / inclusive scan on the DPP crossbar (row_shr 1,2,4,8 inside each row of
// 16 lanes, then row_bcast15 / row_bcast31 across rows; lanes without a source keep `idn`).
// One VALU op per step instead of an LDS-crossbar ds_bpermute round trip.
template <typename Op>
__device__ __forceinline__ int wave_scan_bits(int v, int idn, Op op)
{
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return v;       // inclusive scan; lane 63 holds the total
}
struct OpMinF { __device__ int operator()(int a, int b) const { return __float_as_int(fminf(__int_as_float(a), __int_as_float(b))); } };
struct OpMaxF { __device__ int operator()(int a, int b) const { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); } };
struct OpAddI { __device__ int operator()(int a, int b) const { return a + b; } };

__device__ __forceinline__ float wave_min(float v)
{
    const int r = wave_scan_bits(__float_as_int(v), __float_as_int(__builtin_inff()), OpMinF());
    return __int_as_float(__builtin_amdgcn_readlane(r, 63));
}
__device__ __forceinline__ float wave_max(float v)
{
    const int r = wave_scan_bits(__float_as_int(v), __float_as_int(-__builtin_inff()), OpMaxF());
    return __int_as_float(__builtin_amdgcn_readlane(r, 63));
}
__device__ __forceinline__ int wave_sum_i(int v)
{
    return __builtin_amdgcn_readlane(wave_scan_bits(v, 0, OpAddI()), 63);
}
__device__ __forceinline__ int wave_incl_scan_i(int v) { return wave_scan_bits(v, 0, OpAddI()); }

// histogram slot of logical bin b of an SB-bin histogram: one pad word per BPL = SB/64 bins,
// so that neighbouring bins sit in neighbouring banks (atomics) AND the scan, where lane L
// reads bins BPL*L .. BPL*L+BPL-1, is conflict-free (((BPL+1) L + e) mod 32 distinct over L).
template <int SB> struct SelGeom {
    static constexpr int BPL = SB / 64;                    // bins per lane in the scan
    static constexpr int SLOTS = SB + 64 + 16;             // + pads + the dummy slot, padded to 16
    static_assert(SB % 128 == 0, "bins per lane must be even");
    __device__ static __forceinline__ int slot(int b) { return b + b / BPL; }
};
constexpr int SEL_SLOTS = SelGeom<SEL_BINS>::SLOTS;

struct SelectResult { float value; int cnt_le; float next; };

// k-th smallest (0-based) of a row of n <= 64 * 4 * V4 floats -- exact.  The row is read
// ONCE with V4 back-to-back 16-byte loads per lane (element 256 q + 4 lane + e) and then
// lives in registers; every later pass (range, histogram, gather, next-greater) is
// register + LDS only.  Iterative narrowing: histogram the active value range into
// SEL_BINS linear bins, descend into the bin holding rank k, until it holds <= 64
// elements, which are ranked directly.  If want_next, also returns #(v <= result) and
// min{v > result} (+inf if none).
template <int NV, int SB = SEL_BINS>
__device__ __forceinline__ SelectResult wave_select_regs(const float (&x)[NV], int k, unsigned *hist, float *cand,
                                                         unsigned *counter, int lane, bool want_next)
{
    using SG = SelGeom<SB>;
    constexpr int BPL = SG::BPL;
    const float INF = __builtin_inff();
    float mn = INF, mx = -INF;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        mn = fminf(mn, x[t]);
        mx = fmaxf(mx, (x[t] == INF) ? -INF : x[t]);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    int below = 0;           // elements strictly below the active range [mn, mx]
    float result = mn;
    for (int iter = 0; iter < 64; ++iter) {
        if (!(mn < mx)) { result = mn; break; }
        const float scale = (float)SB / (mx - mn);
        for (int b = lane * 4; b < SG::SLOTS; b += 256)
            *reinterpret_cast<uint4 *>(hist + b) = make_uint4(0, 0, 0, 0);
        if (lane == 0) *counter = 0u;
        wave_lds_fence();
        // Histogram, branch-free: elements outside the active range (the +inf pads, and in later
        // narrowing rounds everything outside [mn, mx]) are routed to a dummy slot, so the NV
        // LDS atomics of a lane issue back to back.
        int bins[NV];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            int b = (int)((x[t] - mn) * scale);
            b = b > SB - 1 ? SB - 1 : b;
            const bool in = x[t] >= mn && x[t] <= mx;
            bins[t] = in ? b : SB;                // SB = dummy
        }
#pragma unroll
        for (int t = 0; t < NV; ++t)
            __hip_atomic_fetch_add(&hist[SG::slot(bins[t])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        wave_lds_fence();
        // scan: lane owns bins [BPL*lane, BPL*lane+BPL)
        int hv[BPL];
        int lsum = 0;
#pragma unroll
        for (int e = 0; e < BPL; ++e) { hv[e] = (int)hist[(BPL + 1) * lane + e]; lsum += hv[e]; }
        const int incl = wave_incl_scan_i(lsum);
        const int target = k - below;                 // rank inside the active set
        const unsigned long long m = __ballot(incl > target);
        const int L = __ffsll((long long)m) - 1;      // first lane whose inclusive sum exceeds target
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
        // wave-uniform (SGPR) copies of lane L's findings
        const int binsel = __builtin_amdgcn_readlane(binsel_v, L);
        const int cum = __builtin_amdgcn_readlane(cum_v, L);
        const int cnt = __builtin_amdgcn_readlane(cnt_v, L);
        if (cnt <= 64) {
            // append the bin's elements (<= 64) to cand[] (order irrelevant), rank, pick.  Only
            // a handful of the NV register slots hold a hit in any lane: a wave-uniform ballot
            // skips the others.
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const bool hit = bins[t] == binsel;
                if (__ballot(hit) != 0ull) {
                    if (hit) {
                        unsigned pos = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        cand[pos & 63u] = x[t];
                    }
                }
            }
            wave_lds_fence();
            const float mine = (lane < cnt) ? cand[lane] : INF;
            int rank = 0;
#pragma unroll 1
            for (int t = 0; t < cnt; ++t) {      // cnt is an SGPR: scalar loop, no LDS traffic
                const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), t));
                rank += (o < mine || (o == mine && t < lane)) ? 1 : 0;
            }
            const int want = target - cum;
            const unsigned long long hitm = __ballot(lane < cnt && rank == want);
            const int src = __ffsll((long long)hitm) - 1;
            result = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), src));
            wave_lds_fence();
            break;
        }
        // narrow to the bin's own value range and iterate
        float nmn = INF, nmx = -INF;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if (bins[t] == binsel) { nmn = fminf(nmn, x[t]); nmx = fmaxf(nmx, x[t]); }
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
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            cle += (x[t] <= result) ? 1 : 0;
            nx = fminf(nx, (x[t] > result) ? x[t] : INF);
        }
        res.cnt_le = wave_sum_i(cle);
        res.next = wave_min(nx);
    }
    return res;
}

// largest f32 x with sqrtf(x) <= eps (inclusive) or sqrtf(x) < eps (exclusive); -1 if none.
// Closed form (no search): sqrtf is correctly rounded, so sqrtf(x) <= e  <=>  sqrt(x) < m, or
// sqrt(x) == m and the tie rounds to e (e's significand even), where m is the midpoint of e and
// the next float above it.  m and m*m are exact in f64 (25 x 25 significant bits), hence the
// threshold is the largest float below m*m -- or m*m itself in the tie case.  "< eps" is
// "<= the float just below eps".
__device__ __forceinline__ float d2_threshold(float eps, int inclusive)
{
    if (!(eps >= 0.0f)) return -1.0f;
    if (eps == __builtin_inff()) return eps;
    float e = eps;
    if (!inclusive) {
        if (eps == 0.0f) return -1.0f;
        e = __uint_as_float(__float_as_uint(eps) - 1u);
    }
    if (__float_as_uint(e) >= 0x7f7fffffu) return e;            // FLT_MAX: every finite x qualifies
    const float en = __uint_as_float(__float_as_uint(e) + 1u);
    const double m = ((double)e + (double)en) * 0.5;
    const double m2 = m * m;
    float c = (float)m2;                                        // round to nearest
    if ((double)c >= m2) {
        const bool tie = (double)c == m2 && (__float_as_uint(e) & 1u) == 0u;
        if (!tie) c = __uint_as_float(__float_as_uint(c) - 1u);  // m2 > 0, so c > 0 here
    }
    return c;
}

// ------------------------------------------------------------------------------------
// Fast path of the band kernel's selection: ONE histogram pass, order statistics k and k+1
// together.  The row lives in registers (NV values per lane, cells outside the matrix are
// +inf).  Returns false when the pass cannot decide (more than 64 candidates in the target
// bins, degenerate value range): the caller then runs the generic narrowing selection.
//   * range: integer min / max on the f32 bit patterns (d2 >= +0, so the patterns order like
//     the values); the max runs on bits + 0x00800000, which makes +inf negative;
//   * bin = (x - mn) * scale, monotone in x, so the histogram is a monotone partition and
//     ranks are exact whatever the rounding; three VALU ops per value give the LDS byte
//     address of the bin (sub, mul, cvt, and_or); +inf lands in the top slot, which no finite
//     value reaches;
//   * lane L owns BINS/64 consecutive bins in the scan; the two target bins are then located
//     with one more LDS read by 2 x 16 lanes; their members (<= 64) are gathered with an
//     address compare and ranked directly.
// `hist_addr` = LDS byte address of this wave's zeroed BINS-dword histogram, aligned to its
// size; `cand` = 64 floats of scratch LDS private to the wave.
// ------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;

// compile-time loop: f(std::integral_constant<int, I>()) for I = B .. E-1
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>());
        static_for<B + 1, E>(f);
    }
}
// (lo, hi)[lane LANE] = the two halves of a wave-uniform 64-bit mask, other lanes keep theirs.
// gfx950 needs 2 wait states between a VALU write of an SGPR (the ballot) and a VALU read of it;
// the compiler does not track that hazard into inline asm, hence the s_nop.
template <int LANE>
__device__ __forceinline__ void writelane_mask(unsigned &lo, unsigned &hi, unsigned long long m)
{
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(lo), "+v"(hi) : "s"((unsigned)m), "s"((unsigned)(m >> 32)), "n"(LANE));
}

struct OpMinU { __device__ int operator()(int a, int b) const { return (unsigned)a < (unsigned)b ? a : b; } };
struct OpMaxI { __device__ int operator()(int a, int b) const { return a > b ? a : b; } };

// NEGPAD: pads are negative instead of +inf.  They are skipped by the range pass without the bias
// add, fall into bin 0 (the conversion saturates at 0) and rank below every cell, so the caller
// passes k already raised by the number of pads that take part in the histogram.
template <int NV, int BINS, int COPIES = 1, bool NEGPAD = false>
__device__ __forceinline__ bool wave_select_fast(const float (&x)[NV], int k, bool want_next, unsigned hist_addr,
                                                 float *cand, int lane, float &slo, float &shi,
                                                 bool lane_has_data = true)
{
    // The histogram has NB = BINS / COPIES logical bins of COPIES counters each; a lane adds to
    // copy (lane % COPIES), which spreads the lanes of one atomic over the banks (bank conflicts,
    // not VALU work, dominate the histogram pass).  Bin b = dwords [b COPIES, +COPIES).
    constexpr int DPL = BINS / 64;          // dwords per lane in the scan
    constexpr int NQ = DPL / 4;             // 16-byte pieces per lane
    constexpr int NB = BINS / COPIES;       // logical bins; the top one only ever holds +inf
    constexpr int BPL = NB / 64;            // logical bins per lane
    static_assert(DPL >= 4 && DPL <= 16 && (DPL & (DPL - 1)) == 0, "BINS must be 256, 512 or 1024");
    static_assert(BPL >= 1 && (COPIES & (COPIES - 1)) == 0 && COPIES <= 8, "COPIES must be 1, 2, 4 or 8 with >= 64 bins");
    const float INF = __builtin_inff();
    // ---- value range over the finite cells
    // (measured: starting the NEGPAD range at 0 instead of the row minimum saves the 32 v_min_u32 but
    // leaves a sixth of the bins empty -- 100.5 vs 99.7 ms per 8192 pairs, no gain)
    unsigned mnu = 0xFFFFFFFFu;
    int mxb = (int)0x80000000;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const unsigned b = __float_as_uint(x[t]);
        mnu = b < mnu ? b : mnu;
        const int bb = NEGPAD ? (int)b : (int)(b + 0x00800000u);
        mxb = bb > mxb ? bb : mxb;
    }
    mnu = (unsigned)__builtin_amdgcn_readlane(wave_scan_bits((int)mnu, -1, OpMinU()), 63);
    mxb = __builtin_amdgcn_readlane(wave_scan_bits(mxb, (int)0x80000000, OpMaxI()), 63);
    const float mn = __uint_as_float(mnu);
    const float mx = __uint_as_float(NEGPAD ? (unsigned)mxb : (unsigned)mxb - 0x00800000u);
    if (mxb < 0) return false;                       // no finite cell at all
    if (NEGPAD && (int)mnu < 0) return false;        // (only pads: cannot happen for a row of the matrix)
    if (!(mn < mx)) { slo = mn; shi = mn; return true; }   // every finite cell equal
    const float range = mx - mn;
    // y = fma(x, scale4, off4) is monotone in x; the rounding of off4 shifts every y by the same
    // amount, at most 2^-24 * mn * scale4 -- kept below one quarter-bin unit by the guard (the top
    // half bin is spare), so no finite value can reach the top slot or wrap the address mask
    if (!(range >= 1e-30f) || !(range <= 1e30f) || !(mn <= 2048.0f * range)) return false;
    const float scale4 = (4.0f * COPIES * ((float)NB - 1.5f)) * __builtin_amdgcn_rcpf(range);   // byte units
    const float off4 = -(mn * scale4);
    // ---- histogram
    unsigned off[NV];
    unsigned vmask = (unsigned)((NB - 1) * 4 * COPIES);
    asm volatile("" : "+v"(vmask));                 // keep the mask in a VGPR: v_and_or_b32 q, vmask, base
    const unsigned hb = hist_addr | (unsigned)((lane & (COPIES - 1)) << 2);   // this lane's copy
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const float y = __builtin_fmaf(x[t], scale4, off4);
        unsigned q;
        asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(y));        // saturating: +inf -> 0xffffffff, y < 0 -> 0
        off[t] = (q & vmask) | hb;
    }
    // lanes that hold nothing but +inf pads stay out: their counts would all land on the top
    // counter (a same-address pile-up in every atomic), which no rank below ever reads
    if (lane_has_data) {
#pragma unroll
        for (int t = 0; t < NV; ++t)
            __hip_atomic_fetch_add((lds_u32 *)off[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wave_lds_fence();
    // ---- scan: lane owns bins [BPL lane, +BPL); pieces read in a staggered order (conflict-free)
    int lsum = 0;
    {
        const int rot = (NQ > 1) ? ((lane >> (NQ == 2 ? 3 : 2)) & (NQ - 1)) : 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int piece = (q + rot) & (NQ - 1);
            const u32x4 h = *(const lds_u32x4 *)(hist_addr + (unsigned)(lane * DPL + 4 * piece) * 4u);
            lsum += (int)(h.x + h.y) + (int)(h.z + h.w);
        }
    }
    const int incl = wave_incl_scan_i(lsum);
    const int L1 = __ffsll((long long)__ballot(incl > k)) - 1;
    const int L2 = want_next ? __ffsll((long long)__ballot(incl > k + 1)) - 1 : L1;
    if (L1 < 0 || L2 < 0) return false;              // cannot happen (k < n <= finite count + pads)
    const int ex1 = __builtin_amdgcn_readlane(incl - lsum, L1);
    const int ex2 = __builtin_amdgcn_readlane(incl - lsum, L2);
    // second level: lanes 0..15 look at lane L1's bins, lanes 16..31 at lane L2's
    const int e = lane & 15;
    const bool lo16 = lane < 16;
    int c = 0;
    if (lane < 32 && e < BPL) {
        const unsigned ba = hist_addr + (unsigned)(((lo16 ? L1 : L2) * BPL + e) * COPIES) * 4u;
        if constexpr (COPIES == 1) {
            c = (int)*(const lds_u32 *)ba;
        } else if constexpr (COPIES == 2) {
            c = (int)(*(const lds_u32 *)ba + *(const lds_u32 *)(ba + 4u));
        } else {
#pragma unroll
            for (int q = 0; q < COPIES / 4; ++q) {
                const u32x4 h = *(const lds_u32x4 *)(ba + 16u * q);
                c += (int)(h.x + h.y) + (int)(h.z + h.w);
            }
        }
    }
    int P = c;                                        // inclusive prefix inside each row of 16 lanes
    P += __builtin_amdgcn_update_dpp(0, P, 0x111, 0xf, 0xf, false);
    P += __builtin_amdgcn_update_dpp(0, P, 0x112, 0xf, 0xf, false);
    P += __builtin_amdgcn_update_dpp(0, P, 0x114, 0xf, 0xf, false);
    if (BPL > 8) P += __builtin_amdgcn_update_dpp(0, P, 0x118, 0xf, 0xf, false);
    const int l1 = __ffsll((long long)__ballot(lo16 && e < BPL && ex1 + P > k)) - 1;
    if (l1 < 0) return false;
    const int cnt1 = __builtin_amdgcn_readlane(c, l1);
    const int cum1 = ex1 + __builtin_amdgcn_readlane(P, l1) - cnt1;
    const int bin1 = L1 * BPL + l1;
    int bin2 = bin1, ncand = cnt1;
    if (want_next) {
        const int l2 = __ffsll((long long)__ballot(!lo16 && lane < 32 && e < BPL && ex2 + P > k + 1)) - 1;
        if (l2 < 0) return false;
        bin2 = L2 * BPL + (l2 - 16);
        if (bin2 != bin1) ncand += __builtin_amdgcn_readlane(c, l2);    // the bins between are empty
    }
    if (ncand > 64 || bin2 >= NB - 1) return false;
    // ---- gather the members of [bin1, bin2]
    unsigned a1 = hb + 4u * COPIES * (unsigned)bin1;
    const unsigned span = 4u * COPIES * (unsigned)(bin2 - bin1);
    // (NEGPAD: the pads of the lanes that stayed out of the histogram share bin 0's address; give those
    // lanes a target no address matches)
    if (NEGPAD && !lane_has_data) a1 = 0x7fffffffu;
    int n = 0;
    auto put = [&](unsigned long long m, bool hit, float v) {
        if (m != 0ull) {
            if (hit) {
                const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                cand[(n + (int)pos) & 63] = v;
            }
            n += __popcll(m);
        }
    };
    static_assert(NV % 4 == 0, "gather works in groups of 4");
    if (span == 0u) {
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            const bool h0 = off[t] == a1, h1 = off[t + 1] == a1, h2 = off[t + 2] == a1, h3 = off[t + 3] == a1;
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            if ((m0 | m1 | m2 | m3) != 0ull) {       // most groups hold no member of the target bin
                put(m0, h0, x[t]); put(m1, h1, x[t + 1]); put(m2, h2, x[t + 2]); put(m3, h3, x[t + 3]);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            const bool h0 = (off[t] - a1) <= span, h1 = (off[t + 1] - a1) <= span;
            const bool h2 = (off[t + 2] - a1) <= span, h3 = (off[t + 3] - a1) <= span;
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            if ((m0 | m1 | m2 | m3) != 0ull) {
                put(m0, h0, x[t]); put(m1, h1, x[t + 1]); put(m2, h2, x[t + 2]); put(m3, h3, x[t + 3]);
            }
        }
    }
    wave_lds_fence();
    // ---- rank them
    // every lane reads the candidates as LDS broadcasts, four per 16-byte read; slots beyond
    // ncand are padded with +inf first so that no tail test is needed
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
    const int want = k - cum1;
    const int s1 = __ffsll((long long)__ballot(lane < ncand && rank == want)) - 1;
    if (s1 < 0) return false;
    slo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), s1));
    shi = slo;
    if (want_next) {
        const int s2 = __ffsll((long long)__ballot(lane < ncand && rank == want + 1)) - 1;
        if (s2 < 0) return false;
        shi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), s2));
    }
    wave_lds_fence();
    return true;
}

// ------------------------------------------------------------------------------------
// Pivot-filtered variant of the fast path (band kernel, negative pads).  Only the smallest ~10 % of a
// row decide its kappa-percentile, so the histogram need not see the rest: a PIVOT g is chosen such
// that a little more than k + 2 cells lie at or below it, the bins cover [row minimum, g] only, and a
// lane adds a cell to the histogram only if its bin exists -- the LDS atomic runs under an exec mask
// with a fifth of the lanes active (4.3 instead of 10.9 LDS cycles per instruction on gfx950,
// scripts/ubench/lds_masked_add.hip) and the pass over the row for its maximum disappears.
//   * pivot: a lane's minimum over its NV consecutive positions is about the 1/(NV+1) quantile of
//     the row, the largest of the 64 / GRP group minima (GRP neighbouring lanes = one group; only
//     groups without pads take part) sits near the (ln 64) / (NV GRP) quantile; g = that maximum
//     pushed up by `delta` of its distance to the row minimum.  Every group then holds a cell <= g;
//     whether k + 2 cells do is CHECKED by the histogram total -- the function returns false when
//     they do not (or on anything else unusual) and the caller falls back to the unfiltered pass.
//   * bin = cvt_u32(fma(x, scale, off)) is monotone in x and a cell takes part iff bin < NB, so the
//     cells that take part are exactly the smallest ones and the histogram is a monotone partition
//     of them: ranks below the total are exact whatever the rounding.
// `hist_addr` = LDS byte address of this wave's zeroed BINS-dword histogram; `cand` = 64 floats.
// ------------------------------------------------------------------------------------
// Pads are +inf (they take no part: k is the plain rank among the cells).
template <int NV, int BINS, int GRP>
__device__ __forceinline__ bool wave_select_pivot(const float (&x)[NV], int k, bool want_next, unsigned hist_addr,
                                                  float *cand, int lane, float &slo, float &shi, bool lane_has_data,
                                                  bool group_full, float delta)
{
    constexpr int NB = BINS;                // logical bins
    constexpr int DPL = BINS / 64;          // dwords per lane in the scan
    constexpr int NQ = DPL / 4;             // 16-byte pieces per lane
    constexpr int BPL = DPL;
    static_assert(DPL >= 4 && DPL <= 16 && (DPL & (DPL - 1)) == 0, "BINS must be 256, 512 or 1024");
    static_assert(GRP == 1 || GRP == 2 || GRP == 4, "groups of 1, 2 or 4 lanes");
    const float INF = __builtin_inff();
    // ---- row minimum and the largest group minimum (unsigned bit patterns: +inf pads are above every cell)
    unsigned mnl = 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const unsigned b = __float_as_uint(x[t]);
        mnl = b < mnl ? b : mnl;
    }
    unsigned gmn = mnl;
    if constexpr (GRP >= 2) {
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)gmn, (int)gmn, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
        gmn = o < gmn ? o : gmn;
    }
    if constexpr (GRP >= 4) {
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)gmn, (int)gmn, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
        gmn = o < gmn ? o : gmn;
    }
    const unsigned mnu = (unsigned)__builtin_amdgcn_readlane(wave_scan_bits((int)mnl, -1, OpMinU()), 63);
    const int mxg = __builtin_amdgcn_readlane(wave_scan_bits(group_full ? (int)gmn : (int)0x80000000, (int)0x80000000, OpMaxI()), 63);
    if (mxg < 0 || (int)mnu < 0) return false;       // no complete group / no cell
    const float mn = __uint_as_float(mnu);
    const float gm = __uint_as_float((unsigned)mxg);
    const float range = __builtin_fmaf(delta, gm - mn, gm) - mn;
    if (!(range >= 1e-30f) || !(range <= 1e30f) || !(mn <= 2048.0f * range)) return false;
    // bin units; the pivot lands in bin NB - 2.  The bin comes out of the fma itself: with 2^23 + 1 folded into
    // the offset the sum is rounded to an integer by the addition (ulp = 1 above 2^23), so the float's bit
    // pattern is MAGIC + bin -- no conversion instruction, and v_fma_f32 issues at twice the rate of v_cvt.
    // offm is exact up to 0.57 (its own rounding + that of mn * scale <= 2^20), so the row minimum and every
    // cell above it land at or above MAGIC: `pattern < MAGIC + NB` alone decides who takes part.  Huge cells
    // and the +inf pads compare above; the lanes that hold nothing but pads get an offset of +inf.
    constexpr unsigned MAGIC = 0x4B000000u;            // 2^23
    const float scale = ((float)NB - 3.0f) * __builtin_amdgcn_rcpf(range);
    const float offm = lane_has_data ? (8388609.0f - mn * scale) : INF;
    // ---- histogram of the cells whose bin exists
    unsigned off[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) off[t] = __float_as_uint(__builtin_fmaf(x[t], scale, offm));
    // The atomics run under exec = [bin < NB].  Written out by hand, four cells per statement: the compiler
    // wraps every exec-masked LDS instruction in a branch of its own (v_cmp, s_and_saveexec, s_cbranch_execz,
    // ..., s_or exec), 32 serialised round trips per row.  exec is put back before the statement ends.
    {
        const unsigned nb = __builtin_amdgcn_readfirstlane(MAGIC + NB);
        const unsigned hb = __builtin_amdgcn_readfirstlane(hist_addr - 4u * MAGIC);      // (mod 2^32, like the shift)
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
    // ---- scan: lane owns bins [BPL lane, +BPL); pieces read in a staggered order (conflict-free)
    int lsum = 0;
    {
        const int rot = (NQ > 1) ? ((lane >> (NQ == 2 ? 3 : 2)) & (NQ - 1)) : 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int piece = (q + rot) & (NQ - 1);
            const u32x4 h = *(const lds_u32x4 *)(hist_addr + (unsigned)(lane * DPL + 4 * piece) * 4u);
            lsum += (int)(h.x + h.y) + (int)(h.z + h.w);
        }
    }
    const int incl = wave_incl_scan_i(lsum);
    const int L1 = __ffsll((long long)__ballot(incl > k)) - 1;
    const int L2 = want_next ? __ffsll((long long)__ballot(incl > k + 1)) - 1 : L1;
    if (L1 < 0 || L2 < 0) return false;              // fewer than k + 2 cells at or below the pivot
    const int ex1 = __builtin_amdgcn_readlane(incl - lsum, L1);
    const int ex2 = __builtin_amdgcn_readlane(incl - lsum, L2);
    // second level: lanes 0..15 look at lane L1's bins, lanes 16..31 at lane L2's
    const int e = lane & 15;
    const bool lo16 = lane < 16;
    int c = 0;
    if (lane < 32 && e < BPL) c = (int)*(const lds_u32 *)(hist_addr + (unsigned)((lo16 ? L1 : L2) * BPL + e) * 4u);
    int P = c;                                        // inclusive prefix inside each row of 16 lanes
    P += __builtin_amdgcn_update_dpp(0, P, 0x111, 0xf, 0xf, false);
    P += __builtin_amdgcn_update_dpp(0, P, 0x112, 0xf, 0xf, false);
    P += __builtin_amdgcn_update_dpp(0, P, 0x114, 0xf, 0xf, false);
    if (BPL > 8) P += __builtin_amdgcn_update_dpp(0, P, 0x118, 0xf, 0xf, false);
    const int l1 = __ffsll((long long)__ballot(lo16 && e < BPL && ex1 + P > k)) - 1;
    if (l1 < 0) return false;
    const int cnt1 = __builtin_amdgcn_readlane(c, l1);
    const int cum1 = ex1 + __builtin_amdgcn_readlane(P, l1) - cnt1;
    const int bin1 = L1 * BPL + l1;
    int bin2 = bin1, ncand = cnt1;
    if (want_next) {
        const int l2 = __ffsll((long long)__ballot(!lo16 && lane < 32 && e < BPL && ex2 + P > k + 1)) - 1;
        if (l2 < 0) return false;
        bin2 = L2 * BPL + (l2 - 16);
        if (bin2 != bin1) ncand += __builtin_amdgcn_readlane(c, l2);    // the bins between are empty
    }
    if (ncand > 64) return false;
    // ---- Round 5: the two order statistics WITHOUT gathering candidates, five rows in six.  Rank k + 1 in a LATER bin than rank k:
    // s_k is the LARGEST cell of bin1 (every cell of bin1 ranks at or below k) and s_(k+1) the SMALLEST of bin2; both ranks in one
    // bin of exactly two cells: its smallest and its largest.  The cells of bin1 post their maximum, those of bin2 their minimum
    // (complemented: both slots start at 0 and take ds_max_u32; unsigned patterns order like the non-negative values) under exec
    // masks a handful of lanes wide -- no candidate list, no ranking loop, no readlane picks.  Same values as the gather below:
    // they ARE the cells of rank k and k + 1.
#ifdef ACX_MINMAX_SHORTCUT      /* development A/B (scripts/ab_build.sh mm -DACX_MINMAX_SHORTCUT).  A MEASURED NEGATIVE with one row per wave: 92.0 vs 93.7 k
                                   pairs/s at T = 2000, 420 vs 424 k at T = 900 -- the per-wave scalars of the gather below cost SALU slots that overlap the
                                   other waves' VALU work, the extrema pass costs 64 vector compares and two LDS round trips (profiles/r05_wide_class.md);
                                   band2_kernel and ef_rowstat2_kernel, where a half cannot keep scalars, are where it pays */
    if (want_next && (bin2 != bin1 || cnt1 == 2)) {
        lds_u32 *ext = (lds_u32 *)(unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)cand;
        if (lane < 2) ext[lane] = 0u;
        wave_lds_fence();
        const unsigned a1 = MAGIC + (unsigned)bin1, a2 = MAGIC + (unsigned)bin2;
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            bool h1[4], h2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { h1[u] = off[t + u] == a1; h2[u] = off[t + u] == a2; }
            if (__ballot(h1[0] || h1[1] || h1[2] || h1[3] || h2[0] || h2[1] || h2[2] || h2[3]) != 0ull) {       // most groups hold no member
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned xb = __float_as_uint(x[t + u]);
                    if (h1[u]) __hip_atomic_fetch_max(ext + 0, xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (h2[u]) __hip_atomic_fetch_max(ext + 1, ~xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        wave_lds_fence();
        const unsigned e1 = ext[0], e2 = ~ext[1];
        wave_lds_fence();
        slo = __uint_as_float(e1 < e2 ? e1 : e2);
        shi = __uint_as_float(e1 < e2 ? e2 : e1);
        return true;
    }
#endif
    // ---- gather the members of [bin1, bin2]
    // (the slots hold MAGIC + bin; pads and the lanes that stayed out of the histogram hold patterns no bin matches)
    const unsigned a1 = MAGIC + (unsigned)bin1;
    const unsigned span = (unsigned)(bin2 - bin1);
    int n = 0;
    auto put = [&](unsigned long long m, bool hit, float v) {
        if (m != 0ull) {
            if (hit) {
                const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                cand[(n + (int)pos) & 63] = v;
            }
            n += __popcll(m);
        }
    };
    static_assert(NV % 4 == 0, "gather works in groups of 4");
    if (span == 0u) {
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            const bool h0 = off[t] == a1, h1 = off[t + 1] == a1, h2 = off[t + 2] == a1, h3 = off[t + 3] == a1;
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            if ((m0 | m1 | m2 | m3) != 0ull) {       // most groups hold no member of the target bin
                put(m0, h0, x[t]); put(m1, h1, x[t + 1]); put(m2, h2, x[t + 2]); put(m3, h3, x[t + 3]);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            const bool h0 = (off[t] - a1) <= span, h1 = (off[t + 1] - a1) <= span;
            const bool h2 = (off[t + 2] - a1) <= span, h3 = (off[t + 3] - a1) <= span;
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            if ((m0 | m1 | m2 | m3) != 0ull) {
                put(m0, h0, x[t]); put(m1, h1, x[t + 1]); put(m2, h2, x[t + 2]); put(m3, h3, x[t + 3]);
            }
        }
    }
    wave_lds_fence();
    // ---- rank them (LDS broadcasts, four per 16-byte read; slots beyond ncand padded with +inf)
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
    const int want = k - cum1;
    const int s1 = __ffsll((long long)__ballot(lane < ncand && rank == want)) - 1;
    if (s1 < 0) return false;
    slo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), s1));
    shi = slo;
    if (want_next) {
        const int s2 = __ffsll((long long)__ballot(lane < ncand && rank == want + 1)) - 1;
        if (s2 < 0) return false;
        shi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), s2));
    }
    wave_lds_fence();
    return true;
}

// eps from the two order statistics d2_(ilo) <= d2_(ihi) (oracle percentile_f32)
__device__ __forceinline__ float percentile_eps2(float slo, float shi, int pct_mode, int ilo, int ihi,
                                                 float kf, float fl, float ce, float &dlo, float &dhi)
{
    // (slo and shi are wave-uniform: odd lanes root shi, even lanes slo -- one sqrt expansion instead of two)
    const float dboth = __builtin_sqrtf((__lane_id() & 1u) ? shi : slo);
    dlo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dboth), 0));
    dhi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dboth), 1));
    if (!(pct_mode == 0 || pct_mode == 1)) return dlo;
    if (pct_mode == 0 && ihi == ilo) return dlo;
    const float d0 = __fmul_rn(dlo, __fsub_rn(ce, kf));
    const float d1 = __fmul_rn(dhi, __fsub_rn(kf, fl));
    return __fadd_rn(d0, d1);
}

// ------------------------------------------------------------------------------------
// K0b: table of embedded squared norms, built once per (pool, m): for every track, every
// rotation rot = 0..11 and every embedded frame i: xx = tree over m frame norms, each a 12-term
// fmaf chain over the bins in ROTATED order (rotated[c'] = src[(c' - rot) mod 12]) -- the chain
// order is part of the arithmetic spec, so the norm depends on the rotation.  Layout:
// tab[noff[track] + rot * (Memb(track) + NGUARD) + i]; the NGUARD entries behind every row (and the slack in
// front of the table) hold +inf: a band-kernel tile that reaches past the matrix reads them as the norms of its
// nonexistent columns, so those cells come out as +inf -- the pad value -- without a single compare.  A pair only picks two rows of it (query unrotated,
// reference rotated by its OTI, or the other way round).
// ------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256) void normtab_kernel(const float *__restrict__ pool,
                                                      const int64_t *__restrict__ toff,
                                                      const int64_t *__restrict__ noff,
                                                      float *__restrict__ tab, int span)
{
    const int track = blockIdx.x, rot = blockIdx.z;            // (tracks on grid.x: any pool size)
    const int64_t t0 = toff[track];
    const int T = (int)(toff[track + 1] - t0);
    const int Me = T - span;                                   // embedded frames (tau == 1)
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= Me) return;
    const float *f = pool + (t0 + i) * NBIN;
    float s[M];
#pragma unroll
    for (int k = 0; k < M; ++k) {
        float v[NBIN];
#pragma unroll
        for (int c = 0; c < NBIN; ++c) v[c] = f[k * NBIN + c];
        float acc = 0.0f;
        for (int cp = 0; cp < NBIN; ++cp) {
            int c = cp - rot; if (c < 0) c += NBIN;
            float x = v[0];
#pragma unroll
            for (int q = 1; q < NBIN; ++q) x = (c == q) ? v[q] : x;
            acc = fmaf(x, x, acc);
        }
        s[k] = acc;
    }
    tab[noff[track] + (int64_t)rot * (Me + NGUARD) + i] = tree_sum<M>(s);
}

// ------------------------------------------------------------------------------------
// K1': fused band kernel.  One workgroup (8 waves) owns a band of 8 rows of the embedded
// distance matrix; the band is cut into tiles of 64 columns and WAVE w sweeps ceil(ntiles / 8)
// CONSECUTIVE tiles on its own -- no workgroup barrier and no LDS staging of inputs in the sweep:
//   * MFMA operands are plain loads from the rotated frame pool (rotpool_kernel): one 12-byte load
//     per lane per 16-frame block, already in the rotated chain order of the arithmetic spec (wide
//     class: buffer loads, scalar tile offset + one lane offset); the next tile's operands are in
//     flight while the current one is worked on; frames and norms OUTSIDE the matrix are read like
//     all others (pool slack, +inf guard norms): every tile runs the same code and the cells
//     outside the matrix come out as +inf, the pad value of the exchange;
//   * the (8+m-1) x (64+7+m-1) frame Gram is built on the matrix cores
//     (v_mfma_f32_16x16x4_f32, K = 12 in 3 k-steps; row-frame operands stay in registers for
//     the whole band; a tile inherits the 16-frame block it shares with its left neighbour) and
//     parked in the wave's private LDS slab (one 16-byte store per block);
//   * lane c walks the 8 cells (a, c + a), a = 0..7, down one diagonal: m+7 LDS reads give
//     all 8 window sums (doubling-tree subterms are shared between the cells); tile t
//     therefore covers, for band row a, the 64 columns 64 t - 7 + a ...;
//   * distances stay in registers (debug variant WD2: also to the row-major D2 matrix in HBM).
// After the sweep the 8 waves exchange their pieces through LDS so that wave w holds band
// row w completely (32 values per lane) and finishes it on its own (band_row_tail): exact
// percentile selection (wave_select_pivot: one pivot-filtered histogram pass; wave_select_fast /
// wave_select_regs behind it), eps, d2-domain threshold and, in the row pass, the bitmap.
// role 1 (launched first): rows = reference frames, columns = query frames (the transposed
//         problem, same bits) -> the column thresholds.
// role 0: rows = query frames -> the row thresholds and, with both thresholds known while the
//         row is still in registers, the binarised row as a 256-byte bitmap.  D2 is never
//         written for thresholds and its transpose is never materialised.
// ------------------------------------------------------------------------------------
constexpr int BAND = 8;
constexpr int BAND_THREADS = 512;   // 8 waves
constexpr int FROT = 3 * NBIN;      // floats per frame of the rotated frame pool

// Rotated frame pool for the band kernel's MFMA operands (built once per upload): frame f ->
// frot[f][r][cls][kb] = frame[f][cls + 4 ((r + kb) mod 3)], r = 0..2, cls = 0..3, kb = 0..2.
static __global__ void rotpool_kernel(const float *__restrict__ pool, float *__restrict__ frot, int64_t nframes)
{
    // (grid-stride: a launch carries at most 2^32 - 1 work-items per dimension; pools beyond 119 M frames exceed that)
    const int64_t total = nframes * FROT, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {     // one output float each
        const int64_t f = idx / FROT;
        const int e = (int)(idx - f * FROT);
        const int r = e / NBIN, rem = e - r * NBIN, cls = rem / 3, kb = rem - 3 * cls;
        frot[idx] = pool[f * NBIN + cls + 4 * ((r + kb) % 3)];
    }
}

// The same pool for the opt-in f16x2 Gram (acx_serra09_params.arith = ACX_ARITH_F16X2): every bin value as TWO fp16 terms
// x = h1 + h2 (h1 = fp16(x), h2 = fp16(x - h1): 22-23 significant bits, fp16 subnormals are kept by the matrix pipe), and per
// (frame, r, cls) the 8 halfs [h1(b0) h1(b1) h1(b2) h2(b0) h2(b1) h2(b2) h1(b0) h1(b1)], b_j = cls + 4 ((r + j) mod 3): exactly the
// first 8 k-slots of the COLUMN operand of v_mfma_f32_16x16x32_f16 (one 16-byte load per lane and 16-frame block); the ninth
// slot (h1(b2)) and the row operand's arrangement [h1 h1 h1 | h1 h1 h1 | h2 h2 h2] are made from the same 16 bytes in registers.
// Products per bin: x1 y1 + x1 y2 + x2 y1 + x2 y2 (12 of the 16 k-slots a lane has in two instructions), exact in f32, accumulated by the pipe.
constexpr int FH = 12 * 8;         // halfs per frame of the f16 operand pool (192 bytes)
static __global__ void rotpool_f16_kernel(const float *__restrict__ pool, _Float16 *__restrict__ fh, int64_t nframes)
{
    const int64_t total = nframes * 12, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {     // one (frame, r, cls) entry each
        const int64_t f = idx / 12;
        const int e = (int)(idx - f * 12), r = e >> 2, cls = e & 3;
        _Float16 h1[3], h2[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float x = pool[f * NBIN + cls + 4 * ((r + j) % 3)];
            h1[j] = (_Float16)x;
            h2[j] = (_Float16)(x - (float)h1[j]);
        }
        _Float16 *o = fh + idx * 8;
        o[0] = h1[0]; o[1] = h1[1]; o[2] = h1[2]; o[3] = h2[0]; o[4] = h2[1]; o[5] = h2[2]; o[6] = h1[0]; o[7] = h1[1];
    }
}

// Pool decimated by the stack stride tau: frame t of the output track = frame t tau of the input
// track (grid: x = track, y = chunks of 256 floats).
static __global__ void decimate_kernel(const float *__restrict__ in, const int64_t *__restrict__ toff_in,
                                       const int64_t *__restrict__ toff_out, float *__restrict__ out, int tau)
{
    const int track = blockIdx.x;
    const int64_t o0 = toff_out[track];
    const int n = (int)(toff_out[track + 1] - o0) * NBIN;
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e >= n) return;
    const int t = e / NBIN, b = e - t * NBIN;
    out[o0 * NBIN + e] = in[(toff_in[track] + (int64_t)t * tau) * NBIN + b];
}

// Host-side launcher of band_kernel: its own translation unit (acx_band.hip), so that the kernel
// can be compiled with the scheduling strategy that suits it without touching the others.
struct BandLaunch {
    hipStream_t stream;
    const float *frot;              // arith 0: rotated f32 frame pool; arith 1 (f16x2): the f16 operand pool, same pointer slot
    const int64_t *toff;
    const float *normtab;
    const int64_t *noff;
    float *scratch, *thr;
    unsigned long long *bits;
    float kappa;
    int pct_mode, inclusive, oti_target;
};
// role 1 / 0 over B pairs of one size class; false when the stack size m has no instantiation
// (want_eps: also evaluate and store every row's eps -- the debug entry point; the production passes skip it where they can)
// (arith: 0 = the exact f32 Gram, 1 = the opt-in f16x2 Gram, m = 9 only; L.frot then points at the f16 operand pool)
bool launch_band_kernel(const BandLaunch &L, int m, const struct PairDesc *dpd, int B, int maxRows, int maxCols, int role,
                        int write_d2, int want_eps, int arith);

// development builds only (scripts/ab_build.sh ablN -DACX_ABL=N): the band kernel stops behind stage N -- 1 sweep, 2 exchange +
// row read, 3 selection, 4 eps / threshold -- so that instruction counters (rocprofv3 --pmc SQ_INSTS_*) can be read per stage
// (profiles/r04_narrow_classes.md).  `val` keeps the stage's result alive.
#ifdef ACX_ABL
#define ACX_ABL_EXIT(n_, val_) do { if (ACX_ABL == (n_)) { if ((val_) == 1.2345e-30f) thr[0] = 1.0f; return; } } while (0)
#else
#define ACX_ABL_EXIT(n_, val_) do { } while (0)
#endif

#ifdef ACX_TIMING   /* development builds only (scripts/ab_build.sh timing -DACX_TIMING; experiments/phase_timing.py) */
__device__ unsigned long long g_band_clk[32];      // [0, 16): band_kernel (slot 15 = waves); [16, 32): spare
struct StampT { unsigned long long t; int base; };
#define ACX_STAMP(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
        if (lane == 0) atomicAdd(&g_band_clk[tstamp_.base + (slot)], now_ - tstamp_.t); tstamp_.t = now_; } while (0)
#else
#define ACX_STAMP(slot) do { } while (0)
#endif

#ifndef ACX_NARROW_WAVES
#define ACX_NARROW_WAVES 8      /* waves per SIMD of the narrowest class (m <= 9): 8 = four workgroups per CU, 6 = three */
#endif
template <int M, int V4 = 8, int ARITH = 0>
struct BandGeom {
    static constexpr int NRT = (BAND + M - 1 + 15) / 16;            // 16-row MFMA tiles of row frames
    static constexpr int NCT = (64 + BAND - 1 + M - 1 + 15) / 16;   // 16-col MFMA tiles of column frames
    static constexpr int AROWS = 16 * NRT;
    static constexpr int BW = 16 * NCT;
    // S pitch: 16-byte aligned rows; 84 % 32 = 20 keeps the 16-byte tile stores conflict-free.  The narrowest class
    // (rows of <= 505 cells, m <= 9) drops the pad: 8 slabs of 16 x 80 floats are exactly 40 KB, a quarter of the
    // CU's LDS, so that FOUR workgroups share a CU instead of three (the tile stores then collide four ways -- 5
    // stores per tile -- which the extra workgroup in flight more than pays for: that class waits on latencies,
    // not on the LDS pipe, which is 26 % busy; profiles/r04_narrow_classes.md)
#ifndef ACX_MID_WAVES
#define ACX_MID_WAVES 8         /* waves per SIMD of the middle class (rows of <= 1017 cells, m <= 9): packed slabs + ONE operand register set (SINGLE_BV) */
#endif
    // (f16x2 Gram: its four-dword column operands do not fit the middle class's 64 registers -- 5 to 9 spills, 13 % slower than the
    // exact kernel -- so that class keeps three workgroups per CU and both operand sets there)
    static constexpr bool PACKED = M <= 9 && ((V4 <= 2 && ACX_NARROW_WAVES >= 8) || (V4 == 4 && ACX_MID_WAVES >= 8 && ARITH == 0));
    static constexpr int SP = PACKED ? BW : BW + 4;
};

// eps from the selected order statistics (oracle percentile_f32)
__device__ __forceinline__ float percentile_eps(const SelectResult &sr, int pct_mode, int ilo, int ihi,
                                                float kf, float fl, float ce)
{
    if (!(pct_mode == 0 || pct_mode == 1)) return __builtin_sqrtf(sr.value);
    float shi = sr.value;
    if (ihi != ilo && sr.cnt_le <= ihi) shi = sr.next;     // rank ihi is the next distinct value
    const float dlo = __builtin_sqrtf(sr.value), dhi = __builtin_sqrtf(shi);
    if (pct_mode == 0 && ihi == ilo) return dlo;
    const float d0 = __fmul_rn(dlo, __fsub_rn(ce, kf));
    const float d1 = __fmul_rn(dhi, __fsub_rn(kf, fl));
    return __fadd_rn(d0, d1);
}

// Geometry of a complete band row in LDS (the exchange) and of the selection's histograms, by row width:
// a row owner holds NV = 8 / 16 / 32 consecutive slots per lane.
template <int NV>
struct RowGeom {
    static constexpr int LNP = NV + 4;          // floats per owner lane in an exchange row: its NV slots + 16 bytes of pad
    static constexpr int ROWP = 64 * LNP;       // exchange pitch (floats)
    // Fast selection: FBINS counters = FBINS / FCOPIES bins x FCOPIES copies.  (Copies spread
    // same-address atomics -- they paid while one instruction handled 64 NEIGHBOURING columns; with
    // the position-order rows one copy and twice the bins measure best on every workload.)
    // For rows of >= 1024 slots the histogram lives in the upper part of the wave's OWN
    // exchange row, free once the row sits in registers; the shortest rows (2 KB) keep a separate
    // area behind the exchange rows.
    static constexpr int FBINS = NV >= 32 ? 1024 : 512;
    static constexpr int FCOPIES = 1;
    static constexpr bool HIST_IN_ROW = ROWP >= 64 + 2 * FBINS;        // room for an FBINS-aligned block behind the 64 candidate slots
    // Pivot-filtered pass (wave_select_pivot): bins, lanes per pivot group and the pivot's push-up, per row
    // width -- groups of 32 cells put the pivot near the 0.13 ... 0.2 quantile of an i.i.d. row
    static constexpr int PBINS = NV >= 32 ? 512 : 256;                  // (256 / 512 / 1024 measure the same at NV = 32)
    static constexpr int PGRP = NV >= 32 ? 1 : 2;
    static constexpr float PDELTA = NV >= 32 ? 0.04f : (NV >= 16 ? 0.15f : 0.0f);
    static constexpr int GBINS = 32 * NV;                              // bins of the generic (narrowing) selection
    static constexpr int TAIL_FLOATS = BAND * ROWP + (HIST_IN_ROW ? 0 : 8 * FBINS);
    static_assert(SelGeom<GBINS>::SLOTS + 64 + 4 <= ROWP, "generic selection must fit the wave's own exchange row");
    static_assert(HIST_IN_ROW || (BAND * ROWP) % FBINS == 0, "fast histograms must be aligned to their size");
    static_assert(NV == 8 || NV == 16 || NV == 32, "row owners hold 8, 16 or 32 consecutive slots");
    // this wave's fast histogram: inside its own exchange row (the first FBINS-aligned block behind
    // the 64 candidate slots), or in the separate area behind the rows
    __device__ static __forceinline__ int hist_off(int wave)
    {
        return HIST_IN_ROW ? ((wave * ROWP + 64 + FBINS - 1) & ~(FBINS - 1)) : BAND * ROWP + wave * FBINS;
    }
};

// The bitmap step of band_row_tail (row pass only): the row in registers against min(row threshold, column thresholds).
template <int NV, int ROLE, bool TC_VIA_LDS, typename TCG>
__device__ __forceinline__ void band_row_bits(const float (&xr)[NV], const TCG (&tcg)[NV / 4], float thr_row, float *myrow, int lane,
                                              int row, int MB, int cshift, const PairDesc &P, float *__restrict__ thr,
                                              unsigned long long *__restrict__ bits)
{
    using RG = RowGeom<NV>;
    constexpr int CH = NV / 4;
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    float *X = thr + P.offX;
    // ---- rows = query frames (the column thresholds of the pair are already there): binarise the row the
    // wave still holds in registers and emit it as a bitmap.  256 bytes per row instead of 8 KB of f32.
    // A lane owns NV consecutive slots, i.e. NV consecutive bits: R = [d2 <= min(thr_row, thr_col)] is
    // shifted into the lane's own word bit by bit (compare -> carry -> add-with-carry), no cross-lane traffic.
    if (ROLE == 0 && bits) {
        // column thresholds (d2 domain) of the lane's NV slots.  In the wide class, read straight from memory, they
        // are 128 contiguous bytes per lane -- every load instruction would touch 64 different lines, and those eight
        // loads measured 4.5 of the band kernels' 47.8 ms.  So the wave reads the row's thresholds
        // lane-interleaved (granule 64 q + lane: 1 KB of contiguous memory per instruction, issued right after
        // the selection, see above) and turns them round through its own exchange row, which the selection no
        // longer needs: 16-byte stores at the granule's padded place, 16-byte loads of the lane's own slots.
        // (no bounds check: columns -7 .. 64 ntiles + 63 of the threshold arena are inside the pair's arena)
        float tcv[NV];
        if constexpr (TC_VIA_LDS) {
            float *tr = myrow + 4 * lane + 4 * ((4 * lane) / NV);
#pragma unroll
            for (int q = 0; q < CH; ++q) *reinterpret_cast<float4 *>(tr + q * (256 + 4 * (256 / NV))) = make_float4(tcg[q].x, tcg[q].y, tcg[q].z, tcg[q].w);
            wave_lds_fence();
            const float *mine = myrow + lane * RG::LNP;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float4 v = *reinterpret_cast<const float4 *>(mine + 4 * j);
                tcv[4 * j + 0] = v.x; tcv[4 * j + 1] = v.y; tcv[4 * j + 2] = v.z; tcv[4 * j + 3] = v.w;
            }
        } else {
            const float *tc = X + P.pitchT + (lane * NV - cshift);      // (the compiler hoists these loads above the selection)
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const f32x4_u v = *reinterpret_cast<const f32x4_u *>(tc + 4 * j);
                tcv[4 * j + 0] = v.x; tcv[4 * j + 1] = v.y; tcv[4 * j + 2] = v.z; tcv[4 * j + 3] = v.w;
            }
        }
        // slots of this lane whose column exists: t in [lo, hi)
        int lo = cshift - lane * NV, hi = MB + cshift - lane * NV;
        lo = lo < 0 ? 0 : lo;
        hi = hi > NV ? NV : hi;
        unsigned valid = 0u;
        if (hi > lo) valid = (hi - lo >= 32 ? ~0u : ((1u << (hi - lo)) - 1u)) << lo;
        unsigned acc = 0u;
#pragma unroll
        for (int t = NV - 1; t >= 0; --t) {
            float mthr;
            asm("v_min_f32 %0, %1, %2" : "=v"(mthr) : "v"(tcv[t]), "v"(thr_row));
            asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(xr[t]), "v"(mthr) : "vcc");
        }
        acc &= valid;
        // NV < 32: neighbouring lanes complete a dword
        // (the lanes that store -- every second / fourth -- take their neighbours' bits inside their quad: DPP quad_perm
        //  [1,1,3,3] / [1,2,3,3], [2,3,3,3] instead of ds_bpermute round trips)
        if constexpr (NV == 16) {
            acc |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)acc, 0xF5, 0xf, 0xf, false) << 16;
        } else if constexpr (NV == 8) {
            acc |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)acc, 0xF9, 0xf, 0xf, false) << 8;
            acc |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)acc, 0xFE, 0xf, 0xf, false) << 16;
        }
        constexpr int LPD = 32 / NV;                                    // lanes per dword
        unsigned *rowbits = reinterpret_cast<unsigned *>(bits + P.offT + (size_t)row * P.nw);
        const int ndw = 2 * P.nw, d = lane / LPD;
        if ((lane & (LPD - 1)) == 0 && d < ndw) rowbits[d] = acc;
        for (int z = 2 * NV + lane; z < ndw; z += 64) rowbits[z] = 0u;  // words beyond this size class
    }
}

// ------------------------------------------------------------------------------------
// The part of the band pipeline that follows the exchange: wave `wave` holds one complete row of the
// pair's matrix in registers (xr[t] = slot NV lane + t; slot s <-> column s - cshift; slots without a
// column hold +inf) and owns exchange row `wave` of `smem` as scratch.  Exact kappa-percentile of the
// row (pivot-filtered histogram, unfiltered histogram, generic narrowing), eps and the d2-domain
// threshold into the pair's threshold arena, and, when `bits` is given (rows = query frames, the
// column thresholds are there already), the binarised row into the recurrence bitmap: bit = slot (bit b of
// row i is column b - 7 + (i & 7), the band kernel's position order).
// ------------------------------------------------------------------------------------
#ifdef ACX_TIMING
#define ACX_STAMP_PARM , StampT &tstamp_
#define ACX_STAMP_ARG , tstamp_
#else
#define ACX_STAMP_PARM
#define ACX_STAMP_ARG
#endif
template <int NV, int ROLE>
__device__ __forceinline__ void band_row_tail(float (&xr)[NV], float *smem, int wave, int lane, int row, int MA, int MB,
                                              int cshift, const PairDesc &P, float *__restrict__ thr,
                                              unsigned long long *__restrict__ bits, const PctPos &pp, int pct_mode,
                                              int inclusive, int want_eps ACX_STAMP_PARM)
{
    using RG = RowGeom<NV>;
    constexpr int ROWP = RG::ROWP, FBINS = RG::FBINS, FCOPIES = RG::FCOPIES, PBINS = RG::PBINS, PGRP = RG::PGRP, GBINS = RG::GBINS;
    constexpr bool HIST_IN_ROW = RG::HIST_IN_ROW;
    constexpr float PDELTA = RG::PDELTA;
    constexpr int CH = NV / 4;           // 16-byte chunks per lane of a complete row
    constexpr int role = ROLE;
    const float INF = __builtin_inff();
    const int hist_off = RG::hist_off(wave);
    if (row >= MA) return;
    const int n = MB;
    const float kf = pp.kf, fl = pp.fl, ce = pp.ce;
    const int ilo = pp.ilo, ihi = pp.ihi, k = pp.k;
    const bool interp = (pct_mode == 0 || pct_mode == 1);
    float *myrow = smem + wave * ROWP;
    float slo, shi;
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned hist_addr = (unsigned)(uintptr_t)(lds_void *)(smem + hist_off);
    // Small kappa (the default 0.095): the pivot-filtered histogram first (wave_select_pivot); it gives up
    // when fewer than k + 2 cells lie below its pivot, and the unfiltered pass takes over.
    const bool use_pivot = (ihi + 2) * 9 <= n;
    auto zero_hist = [&](int bins) {
        float *h = smem + hist_off;
        for (int q = 0; q < bins / 256; ++q) *reinterpret_cast<float4 *>(h + 256 * q + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
        wave_lds_fence();
    };
    if constexpr (HIST_IN_ROW) zero_hist(use_pivot ? PBINS : FBINS);   // the row has left LDS: part of it becomes the zeroed histogram
    bool done = false;
    const int end_valid = MB + cshift;                                  // slots [cshift, end_valid) are cells
    const bool lane_has_data = lane * NV < end_valid;                   // first slot of the lane is a cell or a low pad
    if (use_pivot) {
        const int g0 = (lane & ~(PGRP - 1)) * NV;                       // first slot of the lane's group
        const bool group_full = g0 >= cshift && g0 + PGRP * NV <= end_valid;
        done = wave_select_pivot<NV, PBINS, PGRP>(xr, k, interp && ihi != ilo, hist_addr, myrow, lane, slo, shi,
                                                  lane_has_data, group_full, PDELTA);
        if (!done) zero_hist(FBINS);
    }
    if (!done)
        done = wave_select_fast<NV, FBINS, FCOPIES>(xr, k, interp && ihi != ilo, hist_addr, myrow, lane, slo, shi, lane_has_data);
    if (!done) {
        unsigned *ghist = reinterpret_cast<unsigned *>(myrow) + 64;
        unsigned *counter = reinterpret_cast<unsigned *>(myrow) + 64 + SelGeom<GBINS>::SLOTS;
        const SelectResult sr = wave_select_regs<NV, GBINS>(xr, k, ghist, myrow, counter, lane, interp);
        slo = sr.value;
        shi = (interp && ihi != ilo && sr.cnt_le <= ihi) ? sr.next : sr.value;     // rank ihi is the next distinct value
    }
    ACX_STAMP(5);        // selection
    ACX_ABL_EXIT(3, slo + shi);
    // (row pass: the column thresholds of the row, lane-interleaved, requested here -- behind the selection, whose
    // registers they would otherwise compete for, and ahead of the eps / threshold arithmetic that covers their
    // latency; see the bitmap step)
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    constexpr bool TC_VIA_LDS = NV >= 32;       // (the narrower classes measure 2 % faster with direct loads)
    f32x4_u tcg[CH];
    if constexpr (TC_VIA_LDS) {
        if (role == 0 && bits) {
            asm volatile("" ::: "memory");
            const float *tc = thr + P.offX + P.pitchT - cshift + 4 * lane;
#pragma unroll
            for (int q = 0; q < CH; ++q) tcg[q] = *reinterpret_cast<const f32x4_u *>(tc + 256 * q);
        }
    }
    // Most rows do not even need the roots.  With interpolation weights w0 = ce - kf and w1 = kf - fl of at least 2^-8 each
    // (host-known, the same for every row of the pass) and a relative gap shi - slo > 2^-12 shi between the two order
    // statistics, eps = fl(fl(dlo w0) + fl(dhi w1)) lies in [dlo, dhi) whatever the three roundings do: (dhi - dlo) / dhi >
    // 2^-13 - 2^-23 > 2^-14, so w0 (dhi - dlo) and w1 (dhi - dlo) exceed 2^-22 dhi, four times what the roundings of the two
    // products, the sum and the two correctly rounded roots can move (each <= 2^-24 relative).  Then thr = slo (see below) and
    // neither sqrtf expansion nor eps is evaluated -- unless the caller wants eps itself (the debug entry point).
    const bool weights_ok = interp && ihi == ilo + 1 && inclusive && fl >= 1.0f && (ce - kf) >= 0.00390625f && (kf - fl) >= 0.00390625f;
    if (!want_eps && weights_ok && shi < INF && (shi - slo) > shi * 0.000244140625f) {
        ACX_STAMP(6);
        ACX_ABL_EXIT(4, slo);
        if (lane == 0) (thr + P.offX)[role ? P.pitchT + row : row] = slo;
        band_row_bits<NV, ROLE, TC_VIA_LDS>(xr, tcg, slo, myrow, lane, row, MB, cshift, P, thr, bits);
        ACX_STAMP(7);
        return;
    }
    float dlo, dhi;
    const float eps = percentile_eps2(slo, shi, pct_mode, ilo, ihi, kf, fl, ce, dlo, dhi);
    // The threshold in the d2 domain only has to separate THIS row's cells the way `sqrtf(d2) <= eps` does.  slo and shi are
    // CONSECUTIVE order statistics of the row (ranks ilo and ilo + 1), so with dlo <= eps < dhi (dlo = sqrtf(slo), dhi =
    // sqrtf(shi): what the interpolation gives unless it collapses onto dhi) a cell qualifies iff d2 <= slo: cells <= slo have
    // sqrtf(d2) <= dlo <= eps, every other cell is >= shi and has sqrtf(d2) >= dhi > eps.  thr = slo then -- no f64 midpoint
    // arithmetic (d2_threshold: ~35 VALU operations, some at the f64 rate, per row and pass).  Anything else -- the
    // interpolation collapsed (eps == dhi: ties, an exact-integer position, adjacent roots), eps outside [dlo, dhi) (pct_mode 1
    // at an exact-integer position), the exclusive comparison -- takes the closed form; the branch is wave-uniform.
    float thr_row;
    if (inclusive && dlo <= eps && eps < dhi) thr_row = slo;
    else thr_row = d2_threshold(eps, inclusive);
    ACX_STAMP(6);        // eps + threshold
    float *X = thr + P.offX;
    if (lane == 0) {
        const int o = role ? P.pitchT + row : row;
        X[o] = thr_row;
        X[P.pitchT + P.pitchD + o] = eps;
    }
    band_row_bits<NV, ROLE, TC_VIA_LDS>(xr, tcg, thr_row, myrow, lane, row, MB, cshift, P, thr, bits);
    ACX_STAMP(7);        // threshold store + bitmap
}

// (short-row variants: 6 waves / SIMD = 3 workgroups per CU; m >= 10 needs two MFMA row tiles and is LDS-limited anyway)
constexpr int band_waves_per_simd(int m, int v4, int arith = 0) { return (v4 <= 2 && m <= 9) ? ACX_NARROW_WAVES : ((v4 <= 4 && m <= 9) ? (arith ? 6 : ACX_MID_WAVES) : 4); }
// WD2: the debug entry point's variant, which also writes the band's distances to HBM (D2, query-major) -- a
// template parameter, not a flag: as a run-time flag hipcc folds it into the per-cell store predicates and every
// tile of the production kernel pays 24 VALU + 40 SALU instructions for stores that never happen.
template <int M, int V4, int ROLE, bool WD2 = false, int ARITH = 0>
__global__ __launch_bounds__(BAND_THREADS, band_waves_per_simd(M, V4, ARITH)) void band_kernel(const float *__restrict__ frot,
                                                            const int64_t *__restrict__ toff,
                                                            const float *__restrict__ normtab,
                                                            const int64_t *__restrict__ noff,
                                                            const PairDesc *__restrict__ pd,
                                                            float *__restrict__ scratch,
                                                            float *__restrict__ thr,
                                                            unsigned long long *__restrict__ bits,
                                                            float kappa, int pct_mode, int inclusive, int oti_target, int want_eps)
{
    constexpr bool write_d2 = WD2;
    using G = BandGeom<M, V4, ARITH>;
    constexpr int role = ROLE;           // 1: rows = reference frames (column thresholds); 0: rows = query frames
    constexpr int NV = 4 * V4;           // values per lane of a complete row
    constexpr int NSTEP = NV / 8;        // tiles per wave
    using RG = RowGeom<NV>;
    constexpr int LNP = RG::LNP, ROWP = RG::ROWP, FBINS = RG::FBINS;
    constexpr bool HIST_IN_ROW = RG::HIST_IN_ROW;
    constexpr int SWEEP_FLOATS = 8 * G::AROWS * G::SP;          // one Gram tile per wave
    constexpr int LDS_FLOATS = SWEEP_FLOATS > RG::TAIL_FLOATS ? SWEEP_FLOATS : RG::TAIL_FLOATS;
    __shared__ __attribute__((aligned(4096))) float smem[LDS_FLOATS];

    const int bid_x = blockIdx.x, bid_y = blockIdx.y;
    const PairDesc P = pd[bid_y];
    const int MA = role ? P.Mr : P.Mq, MB = role ? P.Mq : P.Mr;
    const int TA = role ? P.Tr : P.Tq, TB = role ? P.Tq : P.Tr;
    const int i0 = bid_x * BAND;
    if (i0 >= MA) return;     // block-uniform
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR: everything derived from it is scalar
    const bool rows_are_ref = role == 1;
    const int rota = (rows_are_ref == (oti_target == 0)) ? P.oti : 0;
    const int rotb = (rows_are_ref == (oti_target == 0)) ? 0 : P.oti;
    // embedded norms of the row / column track in this pair's rotation (normtab_kernel)
    const float *nrow = normtab + (role ? P.nr : P.nq) + (int64_t)rota * (MA + NGUARD);
    const float *ncol = normtab + (role ? P.nq : P.nr) + (int64_t)rotb * (MB + NGUARD);
    const float INF = __builtin_inff();
    // Cells outside the matrix are +inf, and they get there by themselves: the norm of a nonexistent column
    // (the table's guard entries) or row (below) is +inf, so d2 = (xx - 2 xy) + yy is.  The bit pattern of
    // +inf is above every finite distance: the selection's unsigned minimum never picks it and the
    // pivot-filtered histogram leaves it out.
    const float PADV = INF;

    // ---- MFMA operands come straight from the rotated frame pool (frot, see rotpool_kernel):
    // frame f holds, for each rotation r = 0..2 and residue class cls = 0..3, the three bins
    // cls + 4 ((r + kb) mod 3), kb = 0..2, contiguously.  The lane that feeds k-position lk of
    // the MFMA chain needs rotated bin 4 kb + lk = source bin (4 kb + lk - rot) mod 12, i.e. with
    // c0 = (lk - rot) mod 12 exactly the triple (r = c0 / 4, cls = c0 % 4): ONE 12-byte load per
    // 16-frame tile, already in chain order.  No LDS staging, no LDS-DMA, no barrier.
    const int lr = lane & 15, lk = lane >> 4;
    int c0a = lk - rota; if (c0a < 0) c0a += NBIN;
    int c0b = lk - rotb; if (c0b < 0) c0b += NBIN;
    // ARITH 0: floats of the rotated f32 pool (FROT per frame, a triple per (r, cls)); ARITH 1: halfs of the f16 operand pool (FH per
    // frame, 8 halfs per (r, cls) entry) -- `frot` then points at that pool.  PB = bytes per frame, offA / offB = byte offset of the
    // lane's (r, cls) entry inside a frame.
    constexpr unsigned PB = ARITH ? (unsigned)FH * 2u : (unsigned)FROT * 4u;
    const unsigned offA = ARITH ? (unsigned)(((c0a >> 2) * 4 + (c0a & 3)) * 16) : (unsigned)(((c0a >> 2) * NBIN + (c0a & 3) * 3) * 4);
    const unsigned offB = ARITH ? (unsigned)(((c0b >> 2) * 4 + (c0b & 3)) * 16) : (unsigned)(((c0b >> 2) * NBIN + (c0b & 3) * 3) * 4);
    const char *pool_b = reinterpret_cast<const char *>(frot);
    const char *fra = pool_b + (role ? P.fr : P.fq) * (int64_t)PB + offA;
    // column-frame operands go through a buffer descriptor: scalar base + scalar offset (the tile) + one 32-bit
    // lane offset -- no 64-bit VALU address arithmetic in the tile loop.  The descriptor starts 8 frames before
    // the track (a tile reaches back 7 frames: pool slack / the neighbouring track), no range check.
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void *)(pool_b + ((role ? P.fq : P.fr) - 8) * (int64_t)PB), 0, -1, 0x00020000);
    const unsigned voffB = offB + (unsigned)lr * PB;              // this lane's byte offset inside a 16-frame block
    const char *frb = pool_b + (role ? P.fq : P.fr) * (int64_t)PB + offB;   // (short-row classes: per-lane pointer, plain global loads)
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    typedef f32x3 f32x3_u __attribute__((aligned(4)));
    typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
    typedef u32x3 u32x3_u __attribute__((aligned(4)));
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    // row-frame operands, resident for the whole band.  ARITH 0: three floats (the k-steps of the f32 chain).  ARITH 1: the row
    // arrangement [h1 h1 h1 | h1 h1 h1 | h2 h2 h2] of the entry's first three dwords w0 = (h1b0, h1b1), w1 = (h1b2, h2b0),
    // w2 = (h2b1, h2b2): slots 0-7 in ah8, slot 8 in ah9 (the other slots of that second instruction are zero).
    float areg[G::NRT][3];
    f16x8 ah8[G::NRT], ah9[G::NRT];
#pragma unroll
    for (int ta = 0; ta < G::NRT; ++ta) {
        int f = i0 + 16 * ta + lr;
        f = f > TA - 1 ? TA - 1 : f;           // rows beyond the matrix are masked below
        if constexpr (ARITH == 0) {
            const f32x3 v = *reinterpret_cast<const f32x3_u *>(fra + (size_t)f * PB);
            areg[ta][0] = v.x; areg[ta][1] = v.y; areg[ta][2] = v.z;
        } else {
            const u32x3 w = *reinterpret_cast<const u32x3_u *>(fra + (size_t)f * PB);
            const u32x4v a = {w.x, (w.y & 0xffffu) | (w.x << 16), (w.x >> 16) | (w.y << 16), (w.y >> 16) | (w.z << 16)};
            const u32x4v a9 = {(w.z >> 16) | (w.y & 0xffff0000u), w.z, 0u, 0u};       // slots 8-11: h2b2 | h2b0 h2b1 h2b2
            ah8[ta] = __builtin_bit_cast(f16x8, a);
            ah9[ta] = __builtin_bit_cast(f16x8, a9);
        }
    }
    float xrow[BAND];
#pragma unroll
    for (int a = 0; a < BAND; ++a) xrow[a] = (i0 + a < MA) ? nrow[i0 + a] : INF;       // rows past the matrix: +inf cells
    float *Sw = smem + wave * (G::AROWS * G::SP);               // this wave's Gram tile, [row frame][column frame]
    const PctPos pp = role ? P.pos_q : P.pos_r;                 // rows of MB cells (worked out on the host)
#ifdef ACX_TIMING
    StampT tstamp_{__builtin_readcyclecounter(), 0};
    if (lane == 0) atomicAdd(&g_band_clk[15], 1ull);
#endif

    const int ntiles = (MB + BAND - 1 + 63) / 64;      // <= NV by dispatch
    typedef unsigned BvT[G::NCT][ARITH ? 4 : 3];       // column-frame operands of a tile: dwords per 16-frame block and lane
    typedef f32x4 AccT[G::NRT][G::NCT];
    // column-frame operands of a tile (frames 64 tile - 7 ... + BW)
    // (tb0: first 16-frame block wanted -- a wave's second and later tiles inherit their first HB
    // blocks from the tile before, see the sweep)
    // (edge tiles read frames before / behind the track -- a neighbouring track of the pool or the zeroed slack at
    // its ends, acx.hip POOL_SLACK: finite values, and the cells they feed get a norm of +inf)
    auto load_operands = [&](int tile, BvT &bv, auto tb0_tag) {
        constexpr int tb0 = decltype(tb0_tag)::value;
        const int base = 64 * tile - (BAND - 1);
        if constexpr (V4 >= 8) {
            // wide rows (four tiles per wave): buffer loads, scalar tile offset + 32-bit lane offset (-1.7 % at T = 2000)
            const unsigned so = (unsigned)(base + 8) * PB;                        // wave-uniform byte offset of the tile
#pragma unroll
            for (int tb = tb0; tb < G::NCT; ++tb) {
                if constexpr (ARITH == 0) {
                    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rsB, voffB, so + 16u * PB * tb, 0);
                    bv[tb][0] = v.x; bv[tb][1] = v.y; bv[tb][2] = v.z;
                } else {
                    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB, so + 16u * PB * tb, 0);
                    bv[tb][0] = v.x; bv[tb][1] = v.y; bv[tb][2] = v.z; bv[tb][3] = v.w;
                }
            }
        } else {
            // short rows (one or two tiles per wave): plain global loads measure 4 % faster on the covers80-shaped set
            const char *p = frb + (ptrdiff_t)(base + lr) * (ptrdiff_t)PB;
#pragma unroll
            for (int tb = tb0; tb < G::NCT; ++tb) {
                if constexpr (ARITH == 0) {
                    const u32x3 v = *reinterpret_cast<const u32x3_u *>(p + 16 * PB * tb);
                    bv[tb][0] = v.x; bv[tb][1] = v.y; bv[tb][2] = v.z;
                } else {
                    const u32x4v v = *reinterpret_cast<const u32x4v *>(p + 16 * PB * tb);
                    bv[tb][0] = v.x; bv[tb][1] = v.y; bv[tb][2] = v.z; bv[tb][3] = v.w;
                }
            }
        }
    };
    // embedded column norms of the lane's 8 cells: 8 consecutive norms as two (4-byte aligned) 16-byte loads
    // (columns outside the matrix: the +inf guard entries of the table)
    auto load_norms = [&](int tile, float (&yv)[BAND]) {
        const int base = 64 * tile - (BAND - 1);
        typedef float f32x4n __attribute__((ext_vector_type(4), aligned(4)));
        const f32x4n *p = reinterpret_cast<const f32x4n *>(ncol + base + lane);
        const f32x4n v0 = p[0], v1 = p[1];
        yv[0] = v0.x; yv[1] = v0.y; yv[2] = v0.z; yv[3] = v0.w;
        yv[4] = v1.x; yv[5] = v1.y; yv[6] = v1.z; yv[7] = v1.w;
        static_assert(BAND == 8, "two 16-byte loads cover the band's 8 norms");
    };
    // frame Gram on the matrix cores.  The COLUMN frames are the MFMA's row operand, so a lane ends
    // up with four consecutive column frames of one row frame: one 16-byte LDS store per 16x16 tile
    // (products commute, the k order is unchanged: same bits).  Chains interleaved k-step-major so
    // that no MFMA waits on its predecessor.
    auto gram = [&](const BvT &bv, AccT &acc, auto tb0_tag) {
        constexpr int tb0 = decltype(tb0_tag)::value;
        // (ARITH 1: the zero is opaque to the compiler, so that the first MFMA of a block takes its accumulator from REGISTERS and
        // writes them back.  Round 4 put this in after garbage out of a destination allocated over the dying column operand; round 6
        // measured that the overlap itself is harmless (scripts/ubench/mfma_overlap_probe.hip) and that what bites is a consumer of
        // the result earlier than 5 / 7 wait states (mfma_waitstate_probe.hip) -- scripts/isa_lint.py now checks the shipped code
        // for that; the line stays: it costs nothing and the bits of the default build are pinned by it)
        float zero = 0.0f;
        if constexpr (ARITH != 0) asm volatile("" : "+v"(zero));
#pragma unroll
        for (int ta = 0; ta < G::NRT; ++ta)
#pragma unroll
            for (int tb = tb0; tb < G::NCT; ++tb) acc[ta][tb] = f32x4{zero, zero, zero, zero};
        if constexpr (ARITH == 0) {
#pragma unroll
            for (int kb = 0; kb < 3; ++kb)
#pragma unroll
                for (int ta = 0; ta < G::NRT; ++ta)
#pragma unroll
                    for (int tb = tb0; tb < G::NCT; ++tb)
                        acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(bv[tb][kb]), areg[ta][kb], acc[ta][tb], 0, 0, 0);
        } else {
            // f16x2: slots 0-7 on one v_mfma_f32_16x16x32_f16, slot 8 (h1(b2) of the column frame = the low half of its second
            // dword, h2(b2) of the row frame) on a second one whose other slots are zero: 2 x 16 cycles of the f16 matrix pipe --
            // which runs BESIDE other waves' VALU work -- instead of 3 x 32 cycles of f32 MFMAs that block the SIMD.
            // (Round 4, scripts/ubench/mfma_f16_probe.hip and the mid / wide classes of this kernel: the K = 16 instruction
            // v_mfma_f32_16x16x16_f16 chained behind the K = 32 one gave garbage in kernels with more than one tile per wave
            // although the operands were right.  Round 6, scripts/ubench/mfma_waitstate_probe.hip: another 16-bit MFMA shape may
            // take the K = 32 result as srcC only after 5 wait states (the SAME opcode accumulates back to back), VALU and
            // srcA / srcB readers after 7 -- an earlier consumer sees the destination's stale registers, which is what "returns
            // the operand's bits" was.  One opcode per chain needs no wait at all; scripts/isa_lint.py holds every shipped
            // K = 32 MFMA to the measured distances.)
#pragma unroll
            for (int ta = 0; ta < G::NRT; ++ta)
#pragma unroll
                for (int tb = tb0; tb < G::NCT; ++tb) {
                    const u32x4v b8 = {bv[tb][0], bv[tb][1], bv[tb][2], bv[tb][3]};
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b8), ah8[ta], acc[ta][tb], 0, 0, 0);
                }
#pragma unroll
            for (int ta = 0; ta < G::NRT; ++ta)
#pragma unroll
                for (int tb = tb0; tb < G::NCT; ++tb) {
                    const u32x4v b9 = {bv[tb][1], bv[tb][2], 0u, 0u};                   // slots 8-11: h1b2 | h2b0 h2b1 h2b2 (x2 y1 of b2, then x2 y2)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b9), ah9[ta], acc[ta][tb], 0, 0, 0);
                }
        }
    };
    auto store_gram = [&](const AccT &acc, auto tb0_tag) {
        constexpr int tb0 = decltype(tb0_tag)::value;
#pragma unroll
        for (int ta = 0; ta < G::NRT; ++ta)
#pragma unroll
            for (int tb = tb0; tb < G::NCT; ++tb)
                *reinterpret_cast<f32x4 *>(Sw + (16 * ta + lr) * G::SP + 16 * tb + 4 * lk) = acc[ta][tb];
    };
    // diagonal walk: lane c owns the 8 cells (a, c + a); m + 7 Gram values give all 8 window sums
    auto walk = [&](const float (&sv)[M + BAND - 1], const float (&yv)[BAND], float (&dv)[BAND]) {
#pragma unroll
        for (int a = 0; a < BAND; ++a) {
            const float xy = tree_sum<M>(sv + a);
            // query-side norm first: (xx - 2xy) + yy.  2 * xy is exact, so the fused multiply-add
            // rounds once exactly where the spec's subtraction does.
            float t3 = ROLE ? (__builtin_fmaf(-2.0f, xy, yv[a]) + xrow[a]) : (__builtin_fmaf(-2.0f, xy, xrow[a]) + yv[a]);
            if (!(t3 > 0.0f)) t3 = 0.0f;
            dv[a] = t3;
        }
    };

    const int pitchD = P.pitchD;
    float *D = scratch + P.offD + (size_t)i0 * pitchD;
    float xv[BAND][NSTEP];      // (every element is written below: cells of a tile, or +inf for a tile past the row)

    auto keep = [&](auto st_tag, int tile, const float (&dv)[BAND]) {     // cells -> xv (+ debug D2)
        constexpr int st = decltype(st_tag)::value;
#pragma unroll
        for (int a = 0; a < BAND; ++a) xv[a][st] = dv[a];   // (cells outside the matrix are +inf already)
        if constexpr (write_d2) {
            const int j0 = 64 * tile - (BAND - 1) + lane;   // column of the lane's first cell
#pragma unroll
            for (int a = 0; a < BAND; ++a) {
                const int j = j0 + a;
                if (i0 + a < MA && j >= 0 && j < pitchD) D[a * pitchD + j] = dv[a];
            }
        }
    };

    // ---- sweep: wave w takes CONSECUTIVE tiles w cpw .. w cpw + cpw - 1 (cpw = ceil(ntiles / 8)); the
    // operands of the next tile are in flight (plain global loads, L2 / L1 resident) while the current
    // one is worked on.  Consecutive tiles overlap by the halo: the last HB 16-frame blocks of one
    // tile's Gram ARE the first HB of the next, so from its second tile on a wave stores them again from
    // registers at their new place and multiplies only the four new blocks -- 12 instead of 15 MFMAs
    // per tile at m = 9 (an f32 MFMA blocks the SIMD for 32 cycles), and four operand loads instead of five.
    constexpr int HB = G::NCT - 4;                       // halo blocks shared with the neighbouring tile
    const int cpw = (ntiles + 7) >> 3;                   // tiles per wave (<= NSTEP)
    // tile of step st: a computed one, or (st >= cpw) one of the tiles past all data that this wave pads
    auto tile_of = [&](int st) { return st < cpw ? wave * cpw + st : 8 * cpw + wave * (NSTEP - cpw) + (st - cpw); };
    // Operand registers: two sets (the next tile's loads are issued BEFORE the current tile's MFMAs) -- or, SINGLE_BV, one set that
    // the next tile's loads refill right behind the MFMAs that consumed it (their latency still overlaps the tile's LDS round trip
    // and walk): 15 registers less, which is what lets the middle class hold four workgroups per CU without spilling.
    constexpr bool SINGLE_BV = G::PACKED && V4 == 4;
    BvT bvbuf[SINGLE_BV ? 1 : 2];
    f32x4 halo[G::NRT][HB];
    if (wave * cpw < ntiles) load_operands(wave * cpw, bvbuf[0], std::integral_constant<int, 0>());
    // FULL (the wide class, when every one of the wave's NSTEP tiles exists -- all waves of a band at T = 2000): the same steps without
    // their wave-uniform tests, i.e. ONE basic block for the whole sweep of the wave instead of one per tile
    auto sweep = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    static_for<0, NSTEP>([&](auto st_tag) {
        constexpr int st = decltype(st_tag)::value;
        constexpr int tb0 = st == 0 ? 0 : HB;            // first block this step computes
        const int tile = tile_of(st);
        if (FULL || (st < cpw && tile < ntiles)) {      // wave-uniform
            float yv[BAND];
            load_norms(tile, yv);
            if constexpr (!SINGLE_BV) {
                if (st + 1 < NSTEP && (FULL || (st + 1 < cpw && tile + 1 < ntiles)))
                    load_operands(tile + 1, bvbuf[(st + 1) & 1], std::integral_constant<int, HB>());
            }
            if constexpr (st > 0) {
                // inherit the halo: block 4 + h of the previous tile is block h of this one (still in registers)
#pragma unroll
                for (int ta = 0; ta < G::NRT; ++ta)
#pragma unroll
                    for (int h = 0; h < HB; ++h)
                        *reinterpret_cast<f32x4 *>(Sw + (16 * ta + lr) * G::SP + 16 * h + 4 * lk) = halo[ta][h];
            }
            {
                AccT acc;
                gram(bvbuf[SINGLE_BV ? 0 : (st & 1)], acc, std::integral_constant<int, tb0>());
                if constexpr (SINGLE_BV) {
                    if (st + 1 < NSTEP && (FULL || (st + 1 < cpw && tile + 1 < ntiles)))
                        load_operands(tile + 1, bvbuf[0], std::integral_constant<int, HB>());
                }
                store_gram(acc, std::integral_constant<int, tb0>());
#pragma unroll
                for (int ta = 0; ta < G::NRT; ++ta)
#pragma unroll
                    for (int h = 0; h < HB; ++h) halo[ta][h] = acc[ta][4 + h];
            }
            wave_lds_fence();
            float sv[M + BAND - 1], dv[BAND];
#pragma unroll
            for (int u = 0; u < M + BAND - 1; ++u) sv[u] = Sw[u * G::SP + lane + u];
            walk(sv, yv, dv);
            keep(st_tag, tile, dv);
            wave_lds_fence();
        } else {
#pragma unroll
            for (int a = 0; a < BAND; ++a) xv[a][st] = PADV;
        }
    });
    };
    if constexpr (V4 >= 8 && !WD2) {
        if (cpw == NSTEP && tile_of(NSTEP - 1) < ntiles) sweep(std::true_type());      // (wave-uniform)
        else sweep(std::false_type());
    } else sweep(std::false_type());
    if constexpr (ARITH != 0) {
        // (the row operands outlive every MFMA that reads them: an empty statement that also takes the last tile's cells cannot
        // be scheduled before the last Gram, so no MFMA destination can be allocated over ah8 / ah9)
#pragma unroll
        for (int ta = 0; ta < G::NRT; ++ta) asm volatile("" :: "v"(ah8[ta]), "v"(ah9[ta]), "v"(xv[0][NSTEP - 1]), "v"(xv[BAND - 1][0]));
    }
    // debug / v1 consumers: +inf into the pad columns [MB, pitchD) of the band's rows
    if constexpr (write_d2) {
        const int npad = pitchD - MB;
        for (int idx = tid; idx < BAND * npad; idx += BAND_THREADS) {
            const int a = idx / npad, j = MB + idx - a * npad;
            if (i0 + a < MA) D[a * pitchD + j] = INF;
        }
    }
    ACX_STAMP(0);        // sweep
#ifdef ACX_ABL
    { float keep_ = 0.0f;
      for (int a = 0; a < BAND; ++a) for (int st = 0; st < NSTEP; ++st) keep_ += xv[a][st];
      ACX_ABL_EXIT(1, keep_); }
#endif
    __syncthreads();     // all slabs dead -> reuse LDS as the exchange rows + the fast histograms
    ACX_STAMP(1);        // wait B1
    // ---- exchange: row a of the band becomes a row of LDS in POSITION order (position p = 64 tile +
    // lane <-> column p - 7 + a), so that the owner of the row can pick up NV CONSECUTIVE positions
    // per lane.  With that layout one instruction of the selection handles 64 columns that are NV
    // apart: neighbouring columns of real chroma are similar, and 64 neighbours in one LDS atomic
    // would pile onto a few histogram counters (same-address atomics serialise).  Every owner lane's
    // NV positions are followed by 16 bytes of pad: with a lane pitch of NV + 4 floats the owner's
    // 16-byte reads are bank-conflict free at plain immediate offsets (no address arithmetic at all),
    // and the writers' 4-byte stores stay contiguous up to that pad.
    constexpr int CH = NV / 4;           // 16-byte chunks per lane of a complete row
    constexpr int LPT = 64 / NV == 0 ? 1 : 64 / NV;     // lanes of the owner per 64-position tile
    static_assert(NV == 8 || NV == 16 || NV == 32, "row owners hold 8, 16 or 32 consecutive positions");
    {
        const int wl = lane + 4 * (lane / NV);           // position 64 T + lane -> float T * (LPT * LNP) + wl
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            float *dst = smem + tile_of(st) * (LPT * LNP) + wl;
#pragma unroll
            for (int a = 0; a < BAND; ++a) dst[a * ROWP] = xv[a][st];
        }
    }
    // this wave's fast histogram (RowGeom::hist_off): inside its own exchange row or in the separate area
    if constexpr (!HIST_IN_ROW) {   // zero it before the barrier: nobody else touches that area
        float *h = smem + RG::hist_off(wave);
#pragma unroll
        for (int q = 0; q < FBINS / 256; ++q) *reinterpret_cast<float4 *>(h + 256 * q + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    ACX_STAMP(2);        // exchange writes
    __syncthreads();
    ACX_STAMP(3);        // wait B2
    float xr[NV];      // xr[t] = cell at position NV lane + t of band row `wave` (column = position - 7 + wave)
    {
        const float *mine = smem + wave * ROWP + lane * LNP;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(mine + 4 * j);
            xr[4 * j + 0] = v.x; xr[4 * j + 1] = v.y; xr[4 * j + 2] = v.z; xr[4 * j + 3] = v.w;
        }
    }
    // From here on the waves are independent: wave w owns exchange row w (its cells are in
    // registers now) as private scratch, and its own fast histogram.

    // ---- exact percentile selection, thresholds and (role 0) the recurrence bitmap: wave w owns band row w;
    // slot s of the row = position s = column s - 7 + w
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the row has left LDS
    ACX_STAMP(4);        // row read
#ifdef ACX_ABL
    { float keep_ = 0.0f;
      for (int t = 0; t < NV; ++t) keep_ += xr[t];
      ACX_ABL_EXIT(2, keep_); }
#endif
    band_row_tail<NV, ROLE>(xr, smem, wave, lane, i0 + wave, MA, MB, (BAND - 1) - wave, P, thr, role == 0 ? bits : nullptr,
                            pp, pct_mode, inclusive, want_eps ACX_STAMP_ARG);
}

// One DP row for the CPL columns of a lane (descending column order, in place): QA = row i-1,
// QB = row i-2 (overwritten with row i); l1a / l1b / l2a (p1a / p1b / p2a: their penalised
// versions) = the left neighbour lane's Q[i-1][j0-1], Q[i-1][j0-2], Q[i-2][j0-1]; wraw = raw
// recurrence bits of row i, wprev = of row i-1, wleft bit e = R[i][j-1] (Dmax only).
template <bool EQG, bool DMAX, int CPL>
__device__ __forceinline__ void qmax_cells(unsigned wraw, unsigned colmask, unsigned wprev, unsigned wleft,
                                           float l1a, float l1b, float l2a, float p1a, float p1b, float p2a,
                                           const float (&QA)[CPL], float (&QB)[CPL],
                                           const float (&PA)[EQG ? 1 : CPL], float (&PB)[EQG ? 1 : CPL],
                                           float go, float ge, float &best)
{
    const unsigned w = wraw & colmask;
#pragma unroll
    for (int e = CPL - 1; e >= 0; --e) {
        const bool r = (w >> e) & 1u;
        float x3 = 0.0f, x4 = 0.0f;
        if constexpr (DMAX) {
            x3 = ((wprev >> e) & 1u) ? 1.0f : 0.0f;
            x4 = ((wleft >> e) & 1u) ? 1.0f : 0.0f;
        }
        const float c2 = (e >= 1) ? QA[e - 1] : l1a;                          // (i-1, j-1)
        float c3 = (e >= 1) ? QB[e - 1] : l2a;                                // (i-2, j-1)
        float c4 = (e >= 2) ? QA[e - 2] : (e == 1 ? l1a : l1b);               // (i-1, j-2)
        if constexpr (DMAX) { c3 += x3; c4 += x4; }
        const float mx = fmaxf(fmaxf(c2, c3), c4);
        float q;
        if constexpr (EQG) {
            // match: mx + 1; gap: max(mx - g, 0).  mx >= 0, so both are max(mx + t, 0) with t = +1 or -g
            // picked by the recurrence bit: sign-extend the bit, blend the two constants (v_bfe_i32 +
            // v_bfi_b32 instead of and / compare / select), one add, one max -- same f32 operations on
            // the same values as the two-branch form
            const int m = __builtin_amdgcn_sbfe((int)w, e, 1);
            const float t = __int_as_float((m & __float_as_int(1.0f)) | (~m & __float_as_int(-go)));
            q = fmaxf(mx + t, 0.0f);
        } else {
            const float a2 = (e >= 1) ? PA[e - 1] : p1a;
            float a3 = (e >= 1) ? PB[e - 1] : p2a;
            float a4 = (e >= 2) ? PA[e - 2] : (e == 1 ? p1a : p1b);
            if constexpr (DMAX) { a3 += x3; a4 += x4; }
            const float vgap = fmaxf(fmaxf(fmaxf(a2, a3), a4), 0.0f);
            q = r ? (mx + 1.0f) : vgap;
        }
        // Columns 0, 1 and the columns right of the matrix must not count.  For Qmax the masked
        // recurrence bit does it alone: such a cell takes the gap branch, so it is <= a value an
        // existing cell already reported (never a new maximum), it feeds only cells further
        // right, and in columns 0 / 1 its predecessors are all 0.  Dmax adds raw recurrence bits
        // to the predecessors, so there the cell is forced to 0.
        if constexpr (DMAX) { if (!((colmask >> e) & 1u)) q = 0.0f; }
        QB[e] = q;
        // (Dmax: the penalty of a forced-0 cell in columns 0 / 1 follows its raw recurrence bit)
        if constexpr (!EQG) PB[e] = q - ((DMAX ? (((wraw >> e) & 1u) != 0u) : r) ? go : ge);
    }
#pragma unroll
    for (int e = 0; e < CPL; e += 2) best = fmaxf(best, fmaxf(QB[e], QB[e + 1]));
}

// ------------------------------------------------------------------------------------
// K3b: Qmax on the recurrence bitmap (band pipeline).  One wave per pair; lane owns the CPL
// contiguous columns [CPL lane, +CPL) -- CPL = 32 / 16 / 8 for the three size classes (rows of up
// to 2041 / 1017 / 505 cells), so that short pairs still use all 64 lanes; Q rows i-1 / i-2 live
// in registers (updated in place, descending column order); the row's CPL recurrence bits of a
// lane are one funnel shift of two dwords of the row bitmap (the bitmap of row i starts at column
// (i & 7) - 7).
// ------------------------------------------------------------------------------------
template <bool EQG, bool DMAX, int CPL>
__global__ __launch_bounds__(64) void qmax_bits_kernel(const PairDesc *__restrict__ pd,
                                                       const unsigned long long *__restrict__ bits,
                                                       float *__restrict__ out, int out_stride,
                                                       float go, float ge, int dp_start)
{
    const int lane = threadIdx.x;
    const PairDesc P = pd[blockIdx.x];
    int Me = P.Mq, Ne = P.Mr;
    if (dp_start == 3) { Me -= 1; Ne -= 1; }
    const int ndw = 2 * P.nw;                                   // dwords per row
    const unsigned *rows = reinterpret_cast<const unsigned *>(bits + P.offT);
    // columns of this lane that exist (and are >= 2: the first two columns of Q stay 0)
    unsigned colmask = 0u;
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        const int j = CPL * lane + e;
        if (j >= 2 && j < Ne) colmask |= (1u << e);
    }
    float Q1[CPL], Q2[CPL];
    float P1[EQG ? 1 : CPL], P2[EQG ? 1 : CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        Q1[e] = 0.0f; Q2[e] = 0.0f;
        if constexpr (!EQG) { P1[e] = 0.0f; P2[e] = 0.0f; }
    }
    float best = 0.0f;
    const int prev = (lane + 63) & 63;
    const int dw0 = (CPL * lane) >> 5, bit0 = (CPL * lane) & 31;      // first dword / bit of the lane's columns
    const bool has0 = dw0 < ndw, has1 = dw0 + 1 < ndw;

    auto load_row = [&](int i, unsigned &d0, unsigned &d1) {
        d0 = 0u; d1 = 0u;
        if (i < Me) {
            const unsigned *r = rows + (size_t)i * ndw;
            if (has0) d0 = r[dw0];
            if (has1) d1 = r[dw0 + 1];
        }
    };
    auto row_bits = [&](int i, unsigned d0, unsigned d1) {      // recurrence bits of columns [CPL lane, +CPL)
        const int sh = (BAND - 1) - (i & (BAND - 1));           // bit position of column 0 in the row bitmap
        return __builtin_amdgcn_alignbit(d1, d0, bit0 + sh);    // bit0 + sh <= 31; bits >= CPL are never read
    };
    // Dmax (chen17, latefusion_chen.py:68): the (i-2, j-1) predecessor gains R[i-1][j], the
    // (i-1, j-2) predecessor gains R[i][j-1]; wprev = raw bits of row i-1
    unsigned wprev = 0u;
    if constexpr (DMAX) {
        unsigned p0, p1;
        load_row(1, p0, p1);
        wprev = row_bits(1, p0, p1);
        // With distinct gap penalties the gap branch of row 2 / 3 reads Q - gamma of rows 0 / 1,
        // where Q = 0 but gamma follows the recurrence bit; the Dmax increment can lift such a
        // term above 0, so the two rows start at -gamma instead of 0.
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
    // One DP row: QA = row i-1, QB = row i-2 (overwritten with row i)
    auto dp_row = [&](int i, unsigned d0, unsigned d1, float (&QA)[CPL], float (&QB)[CPL],
                      float (&PA)[EQG ? 1 : CPL], float (&PB)[EQG ? 1 : CPL]) {
        const unsigned wraw = row_bits(i, d0, d1);
        float l1a = lane_prev_f(QA[CPL - 1]), l1b = lane_prev_f(QA[CPL - 2]), l2a = lane_prev_f(QB[CPL - 1]);
        float p1a = 0.f, p1b = 0.f, p2a = 0.f;
        if constexpr (!EQG) {
            p1a = lane_prev_f(PA[CPL - 1]); p1b = lane_prev_f(PA[CPL - 2]); p2a = lane_prev_f(PB[CPL - 1]);
        }
        if (lane == 0) { l1a = 0.f; l1b = 0.f; l2a = 0.f; p1a = 0.f; p1b = 0.f; p2a = 0.f; }
        unsigned wleft = 0u;                                    // bit e = R[i][j-1]
        if constexpr (DMAX) {
            unsigned carry = lane_prev_u((wraw >> (CPL - 1)) & 1u);
            if (lane == 0) carry = 0u;
            wleft = (wraw << 1) | carry;
        }
        qmax_cells<EQG, DMAX, CPL>(wraw, colmask, wprev, wleft, l1a, l1b, l2a, p1a, p1b, p2a, QA, QB, PA, PB, go, ge, best);
        if constexpr (DMAX) wprev = wraw;
    };

    unsigned a0, a1, b0, b1, c0, c1, d0, d1;
    load_row(2, a0, a1); load_row(3, b0, b1); load_row(4, c0, c1); load_row(5, d0, d1);
    for (int i = 2; i < Me; i += 4) {
        // rows i .. i+3; row i+S reads QA = row i+S-1, QB = row i+S-2 and overwrites QB
        unsigned n0, n1;
        if (i < Me) { load_row(i + 4, n0, n1); dp_row(i, a0, a1, Q1, Q2, P1, P2); a0 = n0; a1 = n1; }
        if (i + 1 < Me) { load_row(i + 5, n0, n1); dp_row(i + 1, b0, b1, Q2, Q1, P2, P1); b0 = n0; b1 = n1; }
        if (i + 2 < Me) { load_row(i + 6, n0, n1); dp_row(i + 2, c0, c1, Q1, Q2, P1, P2); c0 = n0; c1 = n1; }
        if (i + 3 < Me) { load_row(i + 7, n0, n1); dp_row(i + 3, d0, d1, Q2, Q1, P2, P1); d0 = n0; d1 = n1; }
    }
    best = wave_max(best);
    if (lane == 0) out[(size_t)blockIdx.x * out_stride] = best;
}

// ------------------------------------------------------------------------------------
// K3c: Qmax with the default penalties (gamma_o = gamma_e = 0.5) in PACKED 16-bit integers.  Every Q
// value is a multiple of 0.5 and at most min(M, N) - 2 <= 2039: in half-units it fits int16, and one
// v_pk_* instruction updates two cells.  A lane still owns CPL contiguous columns; register k holds
// the columns k (low half) and k + CPL/2 (high half), so that the (i-1, j-1), (i-2, j-1), (i-1, j-2)
// predecessors of register k are simply registers k-1 / k-2 of the previous rows (the first two are
// stitched from the neighbour lane with one funnel shift each).  A cell is max(mx + t, 0) with
// t = +2 (match) / -1 (gap) half-units = 3 bit - 1, the bits of both halves spread by one shift + and;
// the clamp at 0 is the saturation of an unsigned subtract (v_pk_mad + v_pk_sub_u16 clamp).
// Exactly the arithmetic of qmax_bits_kernel<true, false, CPL>: integers instead of exact floats.
// DMAX (chen17, latefusion_chen.py:68; qmax_bits_kernel<true, true, CPL>): the (i-2, j-1) predecessor gains R[i-1][j] and the
// (i-1, j-2) predecessor R[i][j-1] -- the RAW recurrence bits, + 2 half-units each (two more v_pk_mad per register) --, and a cell
// outside the matrix or in its first two columns is forced to 0 (a per-register mask).  Dmax values stay below 2 min(M, N)
// = 8164 half-units: int16 holds them.
// ------------------------------------------------------------------------------------
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i16x2 as_i16x2(unsigned v) { return __builtin_bit_cast(i16x2, v); }
__device__ __forceinline__ unsigned as_u32(i16x2 v) { return __builtin_bit_cast(unsigned, v); }

template <int CPL, bool DMAX = false>
__global__ __launch_bounds__(64) void qmax_bits_h16_kernel(const PairDesc *__restrict__ pd,
                                                           const unsigned long long *__restrict__ bits,
                                                           float *__restrict__ out, int out_stride, int dp_start)
{
    constexpr int NR = CPL / 2;                                  // packed registers per row
    const int lane = threadIdx.x;
    const PairDesc P = pd[blockIdx.x];
    int Me = P.Mq, Ne = P.Mr;
    if (dp_start == 3) { Me -= 1; Ne -= 1; }
    const int ndw = 2 * P.nw;
    const unsigned *rows = reinterpret_cast<const unsigned *>(bits + P.offT);
    unsigned colmask = 0u;
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        const int j = CPL * lane + e;
        if (j >= 2 && j < Ne) colmask |= (1u << e);
    }
    unsigned Q1[NR], Q2[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) { Q1[k] = 0u; Q2[k] = 0u; }
    i16x2 best = {0, 0};
    const int prev = (lane + 63) & 63;
    const int dw0 = (CPL * lane) >> 5, bit0 = (CPL * lane) & 31;
    const bool has0 = dw0 < ndw, has1 = dw0 + 1 < ndw;
    auto load_row = [&](int i, unsigned &d0, unsigned &d1) {
        d0 = 0u; d1 = 0u;
        if (i < Me) {
            const unsigned *r = rows + (size_t)i * ndw;
            if (has0) d0 = r[dw0];
            if (has1) d1 = r[dw0 + 1];
        }
    };
    const i16x2 three = {3, 3};
    const i16x2 two = {2, 2};
    const u16x2 one_u = {1, 1};
    // A lane's CPL column bits "arranged": bits [0, NR) stay, bits [NR, CPL) move to [16, 16 + NR) -- once per word and row
    // (three instructions; nothing to do at CPL = 32), so that the bits of columns k and k + NR drop into bit 0 of the two
    // halves of register k with one shift and one mask
    auto arrange = [&](unsigned v) -> unsigned {
        if constexpr (NR == 16) return v;
        else return (v & ((1u << NR) - 1u)) | (((v >> NR) & ((1u << NR) - 1u)) << 16);
    };
    auto spread = [&](unsigned va, int k) -> unsigned { return (va >> k) & 0x00010001u; };     // (of an ARRANGED word)
    const unsigned colmask_a = arrange(colmask);
    unsigned cm[DMAX ? NR : 1];                                  // Dmax: 0xffff in the halves whose column exists and is >= 2
    unsigned wprev = 0u;                                         // Dmax: raw recurrence bits of row i - 1
    if constexpr (DMAX) {
#pragma unroll
        for (int k = 0; k < NR; ++k) cm[k] = spread(colmask_a, k) * 0xffffu;
        unsigned p0, p1;
        load_row(1, p0, p1);
        wprev = arrange(__builtin_amdgcn_alignbit(p1, p0, bit0 + ((BAND - 1) - (1 & (BAND - 1)))));
    } else cm[0] = 0u;
    // One DP row: QA = row i-1, QB = row i-2 (overwritten with row i)
    auto dp_row = [&](int i, unsigned d0, unsigned d1, unsigned (&QA)[NR], unsigned (&QB)[NR]) {
        const int sh = (BAND - 1) - (i & (BAND - 1));
        const unsigned wraw = __builtin_amdgcn_alignbit(d1, d0, bit0 + sh);
        const unsigned wraw_a = arrange(wraw);
        const unsigned w = wraw_a & colmask_a;                   // (arranged, like everything `spread` reads)
        unsigned wleft = 0u;                                     // Dmax: bit e = R[i][j - 1]
        if constexpr (DMAX) wleft = arrange((wraw << 1) | lane_prev_u((wraw >> (CPL - 1)) & 1u));
        // neighbour lane's last registers: its columns CPL-1 / CPL-2 are the HIGH halves of registers NR-1 / NR-2
        unsigned nA1 = lane_prev_u(QA[NR - 1]), nA2 = lane_prev_u(QA[NR - 2]);
        unsigned nB1 = lane_prev_u(QB[NR - 1]);
        if (lane == 0) { nA1 = 0u; nA2 = 0u; nB1 = 0u; }
        // register "-1": low half = column -1 (neighbour's high half of NR-1), high half = column NR-1 (own low half of NR-1)
        const unsigned a_m1 = __builtin_amdgcn_alignbit(QA[NR - 1], nA1, 16);
        const unsigned a_m2 = __builtin_amdgcn_alignbit(QA[NR - 2], nA2, 16);
        const unsigned b_m1 = __builtin_amdgcn_alignbit(QB[NR - 1], nB1, 16);
#pragma unroll
        for (int k = NR - 1; k >= 0; --k) {
            const i16x2 c2 = as_i16x2(k >= 1 ? QA[k - 1] : a_m1);                          // (i-1, j-1)
            i16x2 c3 = as_i16x2(k >= 1 ? QB[k - 1] : b_m1);                                // (i-2, j-1)
            i16x2 c4 = as_i16x2(k >= 2 ? QA[k - 2] : (k == 1 ? a_m1 : a_m2));              // (i-1, j-2)
            if constexpr (DMAX) {
                c3 = as_i16x2(spread(wprev, k)) * two + c3;
                c4 = as_i16x2(spread(wleft, k)) * two + c4;
            }
            const i16x2 mx = __builtin_elementwise_max(__builtin_elementwise_max(c2, c3), c4);
            const unsigned sp = spread(w, k);                // recurrence bit of column k -> low half, of column k + NR -> high half
            // max(mx + 3 bit - 1, 0): multiply-add, then an unsigned saturating subtract (all values are >= 0)
            const u16x2 up = __builtin_bit_cast(u16x2, as_i16x2(sp) * three + mx);
            i16x2 q = __builtin_bit_cast(i16x2, __builtin_elementwise_sub_sat(up, one_u));
            if constexpr (DMAX) q = as_i16x2(as_u32(q) & cm[k]);
            QB[k] = as_u32(q);
            best = __builtin_elementwise_max(best, q);
        }
        if constexpr (DMAX) wprev = wraw_a;
    };
    unsigned a0, a1, b0, b1, c0, c1, d0, d1;
    load_row(2, a0, a1); load_row(3, b0, b1); load_row(4, c0, c1); load_row(5, d0, d1);
    for (int i = 2; i < Me; i += 4) {
        unsigned n0, n1;
        if (i < Me) { load_row(i + 4, n0, n1); dp_row(i, a0, a1, Q1, Q2); a0 = n0; a1 = n1; }
        if (i + 1 < Me) { load_row(i + 5, n0, n1); dp_row(i + 1, b0, b1, Q2, Q1); b0 = n0; b1 = n1; }
        if (i + 2 < Me) { load_row(i + 6, n0, n1); dp_row(i + 2, c0, c1, Q1, Q2); c0 = n0; c1 = n1; }
        if (i + 3 < Me) { load_row(i + 7, n0, n1); dp_row(i + 3, d0, d1, Q2, Q1); d0 = n0; d1 = n1; }
    }
    int bh = best.x > best.y ? best.x : best.y;
    for (int o = 32; o >= 1; o >>= 1) {
        const int t = __shfl_xor(bh, o, 64);
        bh = bh > t ? bh : t;
    }
    if (lane == 0) out[(size_t)blockIdx.x * out_stride] = 0.5f * (float)bh;
}

// ------------------------------------------------------------------------------------
// K3d (round 5): the packed-16-bit Qmax / Dmax of K3c for SEVERAL SHORT pairs per wave.  One wave per pair gives a pair of <= 505
// (249) columns 64 lanes of 8 columns: of the 45 instructions of a row 17 do not depend on the columns a lane owns -- and a covers80
// / DA-TACOS call spends 12 % of its time here.  GLQ = 32 (16) lanes own one pair, 16 columns each, so a wave walks the rows of 2 (4)
// pairs at once: 73 instructions per row for all of them.  Same integers as K3c (and as qmax_bits_kernel<true, DMAX, .>): the
// recursion of a pair never looks beyond its own lanes (the neighbour value of a group's first lane is 0, like lane 0's in K3c), a
// pair that runs out of rows before its neighbours idles on zero bits (its cells only decay: no new maximum).
// grid = ceil(B / (64 / GLQ)) waves; pairs of one size class, sorted or not.
// ------------------------------------------------------------------------------------
template <int GLQ, bool DMAX = false>
__global__ __launch_bounds__(64) void qmax_bits_h16_multi_kernel(const PairDesc *__restrict__ pd, int B,
                                                                 const unsigned long long *__restrict__ bits,
                                                                 float *__restrict__ out, int out_stride, int dp_start)
{
    constexpr int CPL = 16, NR = CPL / 2, PPW = 64 / GLQ;
    static_assert(GLQ == 16 || GLQ == 32, "two or four pairs per wave");
    const int lane = threadIdx.x, l = lane & (GLQ - 1);
    const int pair = blockIdx.x * PPW + lane / GLQ;
    const bool valid = pair < B;
    const PairDesc *P = pd + (valid ? pair : B - 1);
    int Me = valid ? P->Mq : 0, Ne = P->Mr;
    if (dp_start == 3) { Me -= 1; Ne -= 1; }
    const int ndw = 2 * P->nw;
    const unsigned *rows = reinterpret_cast<const unsigned *>(bits + P->offT);
    unsigned colmask = 0u;
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        const int j = CPL * l + e;
        if (j >= 2 && j < Ne) colmask |= (1u << e);
    }
    unsigned Q1[NR], Q2[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) { Q1[k] = 0u; Q2[k] = 0u; }
    i16x2 best = {0, 0};
    const int dw0 = (CPL * l) >> 5, bit0 = (CPL * l) & 31;
    const bool has0 = dw0 < ndw, has1 = dw0 + 1 < ndw;
    auto load_row = [&](int i, unsigned &d0, unsigned &d1) {
        d0 = 0u; d1 = 0u;
        if (i < Me) {
            const unsigned *r = rows + (size_t)i * ndw;
            if (has0) d0 = r[dw0];
            if (has1) d1 = r[dw0 + 1];
        }
    };
    const i16x2 three = {3, 3};
    const i16x2 two = {2, 2};
    const u16x2 one_u = {1, 1};
    auto arrange = [&](unsigned v) -> unsigned { return (v & ((1u << NR) - 1u)) | (((v >> NR) & ((1u << NR) - 1u)) << 16); };
    auto spread = [&](unsigned va, int k) -> unsigned { return (va >> k) & 0x00010001u; };
    const unsigned colmask_a = arrange(colmask);
    unsigned cm[DMAX ? NR : 1];
    unsigned wprev = 0u;
    if constexpr (DMAX) {
#pragma unroll
        for (int k = 0; k < NR; ++k) cm[k] = spread(colmask_a, k) * 0xffffu;
        unsigned p0, p1;
        load_row(1, p0, p1);
        wprev = arrange(__builtin_amdgcn_alignbit(p1, p0, bit0 + ((BAND - 1) - (1 & (BAND - 1)))));
    } else cm[0] = 0u;
    const bool first = l == 0;                        // the first lane of a pair's group has no left neighbour
    auto dp_row = [&](int i, unsigned d0, unsigned d1, unsigned (&QA)[NR], unsigned (&QB)[NR]) {
        const int sh = (BAND - 1) - (i & (BAND - 1));
        const unsigned wraw = __builtin_amdgcn_alignbit(d1, d0, bit0 + sh);        // (bit0 + sh <= 16 + 7)
        const unsigned wraw_a = arrange(wraw);
        const unsigned w = wraw_a & colmask_a;
        unsigned wleft = 0u;
        if constexpr (DMAX) {
            unsigned carry = lane_prev_u((wraw >> (CPL - 1)) & 1u);
            carry = first ? 0u : carry;
            wleft = arrange((wraw << 1) | carry);
        }
        unsigned nA1 = lane_prev_u(QA[NR - 1]), nA2 = lane_prev_u(QA[NR - 2]);
        unsigned nB1 = lane_prev_u(QB[NR - 1]);
        if (first) { nA1 = 0u; nA2 = 0u; nB1 = 0u; }
        const unsigned a_m1 = __builtin_amdgcn_alignbit(QA[NR - 1], nA1, 16);
        const unsigned a_m2 = __builtin_amdgcn_alignbit(QA[NR - 2], nA2, 16);
        const unsigned b_m1 = __builtin_amdgcn_alignbit(QB[NR - 1], nB1, 16);
        // (Dmax: a pair that has run out of rows must not take the bits of its LAST row as "row i - 1" of the idle steps behind it --
        // they would lift the first idle row above 0; found by tests/fuzz_serra09.py on a 3-row matrix beside longer ones)
        const unsigned wp = (DMAX && i < Me) ? wprev : 0u;
#pragma unroll
        for (int k = NR - 1; k >= 0; --k) {
            const i16x2 c2 = as_i16x2(k >= 1 ? QA[k - 1] : a_m1);
            i16x2 c3 = as_i16x2(k >= 1 ? QB[k - 1] : b_m1);
            i16x2 c4 = as_i16x2(k >= 2 ? QA[k - 2] : (k == 1 ? a_m1 : a_m2));
            if constexpr (DMAX) {
                c3 = as_i16x2(spread(wp, k)) * two + c3;
                c4 = as_i16x2(spread(wleft, k)) * two + c4;
            }
            const i16x2 mx = __builtin_elementwise_max(__builtin_elementwise_max(c2, c3), c4);
            const unsigned sp = spread(w, k);
            const u16x2 up = __builtin_bit_cast(u16x2, as_i16x2(sp) * three + mx);
            i16x2 q = __builtin_bit_cast(i16x2, __builtin_elementwise_sub_sat(up, one_u));
            if constexpr (DMAX) q = as_i16x2(as_u32(q) & cm[k]);
            QB[k] = as_u32(q);
            best = __builtin_elementwise_max(best, q);
        }
        if constexpr (DMAX) wprev = wraw_a;
    };
    // the longest pair of the wave sets the number of row steps
    int Mmax = Me;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(Mmax, o, 64); Mmax = Mmax > t ? Mmax : t; }
    Mmax = __builtin_amdgcn_readfirstlane(Mmax);
    unsigned a0, a1, b0, b1, c0, c1, d0, d1;
    load_row(2, a0, a1); load_row(3, b0, b1); load_row(4, c0, c1); load_row(5, d0, d1);
    for (int i = 2; i < Mmax; i += 4) {
        unsigned n0, n1;
        if (i < Mmax) { load_row(i + 4, n0, n1); dp_row(i, a0, a1, Q1, Q2); a0 = n0; a1 = n1; }
        if (i + 1 < Mmax) { load_row(i + 5, n0, n1); dp_row(i + 1, b0, b1, Q2, Q1); b0 = n0; b1 = n1; }
        if (i + 2 < Mmax) { load_row(i + 6, n0, n1); dp_row(i + 2, c0, c1, Q1, Q2); c0 = n0; c1 = n1; }
        if (i + 3 < Mmax) { load_row(i + 7, n0, n1); dp_row(i + 3, d0, d1, Q2, Q1); d0 = n0; d1 = n1; }
    }
    int bh = best.x > best.y ? best.x : best.y;
#pragma unroll
    for (int o = GLQ / 2; o >= 1; o >>= 1) {
        const int t = __shfl_xor(bh, o, 64);
        bh = bh > t ? bh : t;
    }
    if (l == 0 && valid) out[(size_t)pair * out_stride] = 0.5f * (float)bh;
}

static __global__ void sqrt_probe_kernel(const float *in, float *out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_sqrtf(in[i]);
}

}  // namespace acx
