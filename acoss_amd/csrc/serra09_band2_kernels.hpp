// Serra09 band kernel for SHORT rows (<= 761 cells, stack size m <= 9): two -- for rows of <= 249 cells four -- matrix rows per wave.
//
// Reference call site: acoss/algorithms/rqa_serra09.py:44-69 -- the pooled tracks that reach essentia's
// ChromaCrossSimilarity per pair hold 150-650 frames on covers80 / DA-TACOS, so THIS is the shape a user
// sees.  Same arithmetic spec and the same bits as band_kernel (serra09_kernels.hpp); what differs is who
// does what (measurements: profiles/r05_narrow_classes.md):
//   * a workgroup is FOUR (two) waves and still owns a band of 8 matrix rows; a wave sweeps two or three CONSECUTIVE
//     64-column tiles (each inherits the 16-frame halo block of the one before: 12 instead of 15 MFMAs, four operand loads
//     instead of five);
//   * after the exchange a wave owns TWO (four) rows: GL = 32 (16) lanes hold one row, NV = 16 or 24 consecutive positions per
//     lane.  The selection's reductions, prefix scans, bin search, ranking and the threshold arithmetic do not depend on
//     how many values a lane holds -- in band_kernel<M, 2> they are 100 of the selection's 157 VALU instructions per row -- and
//     every one of them now serves all the rows of the wave: reductions stop at the group (DPP inside the 16-lane rows, one
//     v_permlane16_swap across the two rows of a half), the histogram, its scan and the candidates exist once per group, and the
//     groups talk through LDS mailboxes in their own exchange rows instead of through scalars;
//   * the two order statistics are the largest / smallest cell of their histogram bins (pair_select_pivot), no candidate list.
// A row the one-pass selection cannot decide (fewer than k + 2 cells below the pivot, a degenerate range, a large kappa) is
// re-laid out through LDS for the whole wave and goes through band_kernel's own fallbacks (wave_select_fast / wave_select_regs):
// same bits.
#pragma once
#include "serra09_kernels.hpp"

namespace acx {

constexpr int B2_NV = 16;                       // positions per lane of a half-wave row: rows of <= 505 cells (8 tiles) ...
constexpr int B2_NV_MID = 24;                   // ... and, the second class, 24: rows of <= 761 cells (12 tiles) -- every covers80 / DA-TACOS length
// geometry of an exchange row by positions per lane: NV + 4 floats per owner lane (16 bytes of pad: conflict-free 16-byte reads at
// lane pitches of 20 and 28 floats alike -- 4 x an odd number), 32 owner lanes
// GL = lanes that hold one row after the exchange: 32 (two rows per wave) or, for rows of <= 249 cells, 16 (FOUR rows per wave)
template <int NV, int GL = 32> struct B2Geom {
    static_assert(NV == 16 || NV == 24, "16 or 24 positions per lane");
    static_assert(GL == 32 || (GL == 16 && NV == 16), "half-wave rows, or quarter-wave rows of 256 positions");
    static constexpr int RPW = 64 / GL;         // rows per wave
    static constexpr int WAVES = BAND / RPW;    // waves of a workgroup: 4 / 2
    static constexpr int LNP = NV + 4;
    static constexpr int ROWP = GL * LNP;       // 640 / 896 / 320 floats
    static constexpr int NSTEP = NV * GL / 64 / WAVES;     // tiles per wave of the sweep: 2 / 3 / 2
    static constexpr int BINS = GL == 32 ? 256 : 128;      // 8 bins per lane of a row's group either way
};
#ifndef ACX_B2_WAVES_PER_SIMD
#define ACX_B2_WAVES_PER_SIMD 7      /* 72 registers, 0-4 spilled; eight (64 registers) spill 10-14: 276 vs 308 (six) vs 332 (seven) Gcells/s at T = 450 */
#endif
#ifndef ACX_B2_MID_WAVES_PER_SIMD
#define ACX_B2_MID_WAVES_PER_SIMD 5  /* the 24-position class: 96 registers, 28.7 KB of exchange rows */
#endif
constexpr int b2_waves_per_simd(int nv) { return nv == 16 ? ACX_B2_WAVES_PER_SIMD : ACX_B2_MID_WAVES_PER_SIMD; }
constexpr int b2_threads(int gl) { return 64 * (BAND / (64 / gl)); }          // 256 (two rows per wave) / 128 (four)

// ---- reductions over the 32 lanes of a half, result in EVERY lane of the half: xor 1, xor 2 inside the quads, mirror inside 8
// and 16 lanes (DPP, one VALU operation each), then the two 16-lane rows of the half trade places (v_permlane16_swap).
template <int GL, typename Op>
__device__ __forceinline__ int group_allreduce(int v, int idn, Op op);
template <typename Op>
__device__ __forceinline__ int half_allreduce(int v, int idn, Op op)
{
    // (`idn` = the operation's identity as the DPP "old" value: every lane has a source here, so it is never used -- but with it
    // hipcc folds the move into the operation, v_min_u32_dpp instead of v_mov + v_mov_dpp + v_min)
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x141, 0xf, 0xf, false));    // row_half_mirror
    v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x140, 0xf, 0xf, false));    // row_mirror
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);   // r[0]: rows 0 0 2 2, r[1]: rows 1 1 3 3
    return op((int)r[0], (int)r[1]);
}
// the same over the GL lanes that hold one row: a half (above), or one 16-lane DPP row (no exchange between rows needed)
template <int GL, typename Op>
__device__ __forceinline__ int group_allreduce(int v, int idn, Op op)
{
    if constexpr (GL == 32) return half_allreduce(v, idn, op);
    else {
        v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0xB1, 0xf, 0xf, false));
        v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x4E, 0xf, 0xf, false));
        v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x141, 0xf, 0xf, false));
        v = op(v, __builtin_amdgcn_update_dpp(idn, v, 0x140, 0xf, 0xf, false));
        return v;
    }
}
// inclusive prefix sum inside each half (lanes without a source add 0)
__device__ __forceinline__ int half_incl_scan_i(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    return v;
}
template <int GL>
__device__ __forceinline__ int group_incl_scan_i(int v)
{
    if constexpr (GL == 32) return half_incl_scan_i(v);
    else {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
        return v;
    }
}

// ------------------------------------------------------------------------------------
// Pivot-filtered one-pass selection (wave_select_pivot, serra09_kernels.hpp) for TWO rows at once: lanes 0-31 hold
// one row, lanes 32-63 the other, NV consecutive positions per lane, pads +inf.  k (0-based rank) is the same for
// both rows (they belong to one pair and one pass).
//
// The two halves work like two independent 32-lane groups that talk through LDS "mailboxes" in their own row's area
// (`mb_addr`: LDS byte address of the half's 64 zeroed mailbox dwords, followed by its zeroed B2Geom::BINS-dword histogram) --
// no per-half scalars, no v_readlane / v_cndmask glue:
//   1. histogram of the cells at or below the pivot (exec-masked ds_add_u32, as wave_select_pivot);
//   2. prefix scan of the lanes' 8-bin sums inside the half; the lane whose range holds rank k (k + 1) posts {lane + 1,
//      exclusive count}; lanes 0-7 (16-23) of the half read lane L1's (L2's) 8 bins, scan them inside their DPP row and the
//      lane whose bin holds the rank posts {bin, cells below the bin, cells in the bin};
//   3. the two order statistics WITHOUT gathering candidates: rank k + 1 lies in a later bin than rank k (five rows of
//      six at 0.4 cells per bin) => s_k is the LARGEST cell of bin1 and s_(k+1) the SMALLEST of bin2; both in one bin that
//      holds exactly two cells => its smallest and its largest.  So the cells of bin1 post their maximum, those of bin2
//      their minimum (exec-masked ds_max_u32 on the patterns / their complements; a handful of lanes), and s_k = min, s_(k+1)
//      = max of the two posted values in either case;
//   4. a bin with three or more cells around the rank (heavy ties, one row in sixteen on i.i.d. data): its <= 32 members are
//      gathered through an LDS counter and ranked directly, as wave_select_pivot does.
// Returns in every lane whether ITS half has both order statistics (slo, shi; shi = slo without want_next); a half
// that has not goes to the caller's fallback.
// ------------------------------------------------------------------------------------
struct B2Mail {            // dword offsets inside a half's mailbox area
    static constexpr int CAND = 0;      // 32 candidate slots (step 4)
    static constexpr int COUNTER = 32;  // their counter
    static constexpr int E = 34;        // [34] max pattern of bin1's cells, [35] max complemented pattern of bin2's cells
    static constexpr int OWN = 36;      // [36, 37] {L1 + 1, excl}, [38, 39] {L2 + 1, excl}
    static constexpr int BIN = 40;      // [40..42] {bin1, below1, count1}, [44..46] {bin2, below2, count2}
    static constexpr int RES = 48;      // [48] s_k, [49] s_(k+1) of step 4
    static constexpr int HIST = 64;
};
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;

template <int NV, int GL = 32>
__device__ __forceinline__ bool pair_select_pivot(const float (&x)[NV], int k, bool want_next, unsigned mb_addr, int lane,
                                                  float &slo, float &shi, bool lane_has_data, bool group_full, bool fine_pivot)
{
    constexpr int NB = B2Geom<NV, GL>::BINS, BPL = NB / GL;      // 8 bins per lane in the scan
    static_assert(NV % 4 == 0, "atomics and the extrema pass go in groups of 4");
    static_assert(BPL == 8, "two 16-byte pieces per lane");
    const float INF = __builtin_inff();
    const int l = lane & (GL - 1);
    const unsigned hist_addr = mb_addr + 4u * B2Mail::HIST;
    // ---- row minimum and the pivot: the largest minimum of a lane whose NV positions are all cells (unsigned patterns: +inf pads
    // above every cell).  A SHORT row has few such lanes -- 8 of them leave 12 % of the row below the pivot where the rank needs
    // 9.5 % + 2 cells --: `fine_pivot` (wave-uniform) takes the largest minimum of a HALF lane instead, twice the groups of half
    // the size: 30 % of a row of 142 cells.  Any pivot gives the same order statistics; it only decides how often the row is sent on.
    unsigned mna = 0xFFFFFFFFu, mnb = 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < NV / 2; ++t) {
        const unsigned a = __float_as_uint(x[t]), b = __float_as_uint(x[NV / 2 + t]);
        mna = a < mna ? a : mna;
        mnb = b < mnb ? b : mnb;
    }
    const unsigned mnl = mna < mnb ? mna : mnb;
    const unsigned pvl = fine_pivot ? (mna < mnb ? mnb : mna) : mnl;
    const unsigned mnu = (unsigned)group_allreduce<GL>((int)mnl, -1, OpMinU());
    const int mxg = group_allreduce<GL>(group_full ? (int)pvl : (int)0x80000000, (int)0x80000000, OpMaxI());
    const float mn = __uint_as_float(mnu);
    const float gm = __uint_as_float((unsigned)mxg);
    const float range = gm - mn;
    // (the same guards as wave_select_pivot; `good` is uniform inside a half)
    const bool good = mxg >= 0 && (int)mnu >= 0 && range >= 1e-30f && range <= 1e30f && mn <= 2048.0f * range;
    constexpr unsigned MAGIC = 0x4B000000u;            // 2^23: the bin is the low part of the binning fma's bit pattern
    const float scale = ((float)NB - 3.0f) * __builtin_amdgcn_rcpf(range);
    const float offm = (lane_has_data && good) ? (8388609.0f - mn * scale) : INF;
    unsigned pat[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) pat[t] = __float_as_uint(__builtin_fmaf(x[t], scale, offm));
    // ---- 1. histogram of the cells whose bin exists: exec-masked LDS atomics, four cells per hand-written statement
    {
        const unsigned nb = __builtin_amdgcn_readfirstlane(MAGIC + NB);
        const unsigned hb = hist_addr - 4u * MAGIC;                        // (mod 2^32, like the shift); per half
        unsigned one = 1u;
        asm volatile("" : "+v"(one));
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
                         : [q0] "v"(pat[t]), [q1] "v"(pat[t + 1]), [q2] "v"(pat[t + 2]), [q3] "v"(pat[t + 3]),
                           [nb] "s"(nb), [hb] "v"(hb), [one] "v"(one)
                         : "memory");
        }
    }
    wave_lds_fence();
    // ---- 2. scan: lane l of a half owns bins [8 l, 8 l + 8) of its half's histogram; the two 16-byte pieces in a staggered
    // order (lanes 8 apart would otherwise meet in one bank group)
    int lsum = 0;
    {
        const int rot = (l >> 3) & 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int piece = (q + rot) & 1;
            const u32x4 h = *(const lds_u32x4 *)(hist_addr + (unsigned)(l * BPL + 4 * piece) * 4u);
            lsum += (int)(h.x + h.y) + (int)(h.z + h.w);
        }
    }
    const int incl = group_incl_scan_i<GL>(lsum);
    const int excl = incl - lsum;
    const int k2 = want_next ? k + 1 : k;
    // the lane whose bins hold the rank posts itself (at most one per half and rank; none: fewer than k + 2 cells below the pivot)
    if (excl <= k && k < incl) *(lds_u32x2 *)(mb_addr + 4u * B2Mail::OWN) = u32x2{(unsigned)l + 1u, (unsigned)excl};
    if (excl <= k2 && k2 < incl) *(lds_u32x2 *)(mb_addr + 4u * (B2Mail::OWN + 2)) = u32x2{(unsigned)l + 1u, (unsigned)excl};
    wave_lds_fence();
    // second level: lanes 0..7 of a half look at the 8 bins of lane L1, lanes 16..23 at those of lane L2 (quarter-wave rows:
    // lanes 0..7 and 8..15 of the group)
    const int e = l & (GL / 2 - 1);
    const bool second = (l & (GL / 2)) != 0;
    const u32x2 own = *(const lds_u32x2 *)(mb_addr + 4u * B2Mail::OWN + (second ? 8u : 0u));
    const int Lx = (int)own.x - 1, exx = (int)own.y;                  // (Lx = -1: nobody posted; the read below then hits the mailbox area)
    const int kk = second ? k2 : k;
    int c = (int)*(const lds_u32 *)(hist_addr + (unsigned)(Lx * (BPL * 4) + (e & 7) * 4));
    c = e < BPL ? c : 0;
    int P = c;                                        // inclusive prefix over the 8 bins
    if constexpr (GL == 32) {                         // (they sit at the start of a 16-lane DPP row: plain row shifts)
        P += __builtin_amdgcn_update_dpp(0, P, 0x111, 0xf, 0xf, false);
        P += __builtin_amdgcn_update_dpp(0, P, 0x112, 0xf, 0xf, false);
        P += __builtin_amdgcn_update_dpp(0, P, 0x114, 0xf, 0xf, false);
    } else {                                          // (two groups of 8 share a DPP row: the second must not see the first)
        int t_ = __builtin_amdgcn_update_dpp(0, P, 0x111, 0xf, 0xf, false); P += e >= 1 ? t_ : 0;
        t_ = __builtin_amdgcn_update_dpp(0, P, 0x112, 0xf, 0xf, false); P += e >= 2 ? t_ : 0;
        t_ = __builtin_amdgcn_update_dpp(0, P, 0x114, 0xf, 0xf, false); P += e >= 4 ? t_ : 0;
    }
    {
        const int below = exx + P - c;                // cells below this lane's bin
        if (e < BPL && Lx >= 0 && below <= kk && kk < below + c) {
            typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
            typedef __attribute__((address_space(3))) u32x3 lds_u32x3;
            *(lds_u32x3 *)(mb_addr + 4u * B2Mail::BIN + (second ? 16u : 0u)) = u32x3{(unsigned)(Lx * BPL + e), (unsigned)below, (unsigned)c};
        }
    }
    wave_lds_fence();
    const u32x4 r1 = *(const lds_u32x4 *)(mb_addr + 4u * B2Mail::BIN), r2 = *(const lds_u32x4 *)(mb_addr + 4u * (B2Mail::BIN + 4));
    const int bin1 = (int)r1.x, below1 = (int)r1.y, cnt1 = (int)r1.z, bin2 = (int)r2.x, cnt2 = (int)r2.z;
    const bool found = good && cnt1 != 0 && cnt2 != 0;               // (zeroed mailbox: nobody posted)
    // ---- 3. the largest cell of bin1, the smallest of bin2 (complemented: the mailbox is zeroed, both are ds_max_u32)
    const unsigned a1 = found ? MAGIC + (unsigned)bin1 : 0u, a2 = found ? MAGIC + (unsigned)bin2 : 0u;    // (0 matches no pattern)
    {
        lds_u32 *e1 = (lds_u32 *)(mb_addr + 4u * B2Mail::E), *e2 = (lds_u32 *)(mb_addr + 4u * (B2Mail::E + 1));
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            bool h1[4], h2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { h1[u] = pat[t + u] == a1; h2[u] = pat[t + u] == a2; }
            if (__ballot(h1[0] || h1[1] || h1[2] || h1[3] || h2[0] || h2[1] || h2[2] || h2[3]) != 0ull) {     // most groups hold no member
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (h1[u]) __hip_atomic_fetch_max(e1, __float_as_uint(x[t + u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (h2[u]) __hip_atomic_fetch_max(e2, ~__float_as_uint(x[t + u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    wave_lds_fence();
    const u32x2 ev = *(const lds_u32x2 *)(mb_addr + 4u * B2Mail::E);
    const unsigned E1 = ev.x, E2 = ~ev.y;
    bool ok;
    if (want_next) {
        ok = found && (bin1 != bin2 || cnt1 == 2);
        slo = __uint_as_float(E1 < E2 ? E1 : E2);
        shi = __uint_as_float(E1 < E2 ? E2 : E1);
    } else {
        const int p = k - below1;                     // position of rank k inside its bin
        ok = found && (p == 0 || p == cnt1 - 1);
        slo = __uint_as_float(p == 0 ? E2 : E1);
        shi = slo;
    }
    // ---- 4. three or more cells in the bin around the rank: gather the members of [bin1, bin2] (<= 32) and rank them
    const int ncand = (bin2 != bin1) ? cnt1 + cnt2 : cnt1;            // (the bins between are empty)
    const bool crowded = found && !ok && ncand <= GL;
    if (__ballot(crowded) != 0ull) {
        const unsigned cand_addr = mb_addr + 4u * B2Mail::CAND;
        lds_u32 *counter = (lds_u32 *)(mb_addr + 4u * B2Mail::COUNTER);
        const unsigned g1 = crowded ? a1 : 0u, span = (unsigned)(bin2 - bin1);
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if ((pat[t] - g1) <= span) {
                const unsigned pos = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                *(lds_f32 *)(cand_addr + 4u * (pos & (unsigned)(GL - 1))) = x[t];
            }
        }
        wave_lds_fence();
        const int nc = crowded ? ncand : 0;
        if (l >= nc) *(lds_f32 *)(cand_addr + 4u * (unsigned)l) = INF;
        wave_lds_fence();
        const float mine = *(const lds_f32 *)(cand_addr + 4u * (unsigned)l);
        int rank = 0;
        for (int t = 0; __ballot(t < nc) != 0ull; t += 4) {
            const f32x4 o = *(const lds_f32x4 *)(cand_addr + 4u * (unsigned)t);
            rank += (o.x < mine || (o.x == mine && t + 0 < l)) ? 1 : 0;
            rank += (o.y < mine || (o.y == mine && t + 1 < l)) ? 1 : 0;
            rank += (o.z < mine || (o.z == mine && t + 2 < l)) ? 1 : 0;
            rank += (o.w < mine || (o.w == mine && t + 3 < l)) ? 1 : 0;
        }
        const int want = k - below1;
        if (l < nc && rank == want) *(lds_f32 *)(mb_addr + 4u * B2Mail::RES) = mine;
        if (l < nc && rank == want + (want_next ? 1 : 0)) *(lds_f32 *)(mb_addr + 4u * (B2Mail::RES + 1)) = mine;
        wave_lds_fence();
        if (crowded) {
            const u32x2 rr = *(const lds_u32x2 *)(mb_addr + 4u * B2Mail::RES);
            slo = __uint_as_float(rr.x); shi = __uint_as_float(rr.y);
            ok = true;          // (ranks want and want + 1 exist among the nc = count of [bin1, bin2] members)
        }
    }
    wave_lds_fence();
    return ok;
}

// ------------------------------------------------------------------------------------
// band2_kernel: see the head of this file.  grid = (bands of 8 rows, pairs), 256 threads.
// ------------------------------------------------------------------------------------
template <int M, int ROLE, bool WD2 = false, int NVT = B2_NV, int GLT = 32>
__global__ __launch_bounds__(b2_threads(GLT), b2_waves_per_simd(NVT)) void band2_kernel(const float *__restrict__ frot,
                                                            const float *__restrict__ normtab,
                                                            const PairDesc *__restrict__ pd,
                                                            float *__restrict__ scratch,
                                                            float *__restrict__ thr,
                                                            unsigned long long *__restrict__ bits,
                                                            int pct_mode, int inclusive, int oti_target, int want_eps)
{
    static_assert(M <= 9, "one 16-row MFMA tile of row frames");
    constexpr bool write_d2 = WD2;
    constexpr int role = ROLE;           // 1: rows = reference frames (column thresholds); 0: rows = query frames
    constexpr int NV = NVT, GL = GLT;
    using BG = B2Geom<NV, GL>;
    constexpr int LNP = BG::LNP, ROWP = BG::ROWP, RPW = BG::RPW, WAVES = BG::WAVES;
    constexpr int NSTEP = BG::NSTEP;               // tiles per wave: 2 (rows of <= 249 / 505 cells) or 3 (<= 761)
    constexpr int B2_LDS_FLOATS = BAND * ROWP;     // the 8 exchange rows: 10 / 20 / 28 KB
    constexpr int NCT = (64 + BAND - 1 + M - 1 + 15) / 16;   // 16-column MFMA blocks of a tile (5)
    constexpr int SP = 16 * NCT + 4;     // Gram slab pitch: 84 % 32 = 20 keeps the 16-byte tile stores conflict-free
    constexpr int LDS_FLOATS = WAVES * 16 * SP > B2_LDS_FLOATS ? WAVES * 16 * SP : B2_LDS_FLOATS;     // NV = 16: 21.5 KB of Gram slabs, six workgroups per CU; NV = 24: 28 KB of exchange rows, five
    __shared__ __attribute__((aligned(4096))) float smem[LDS_FLOATS];

    const PairDesc P = pd[blockIdx.y];
    const int MA = role ? P.Mr : P.Mq, MB = role ? P.Mq : P.Mr;
    const int TA = role ? P.Tr : P.Tq;
    const int i0 = blockIdx.x * BAND;
    if (i0 >= MA) return;     // block-uniform
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool rows_are_ref = role == 1;
    const int rota = (rows_are_ref == (oti_target == 0)) ? P.oti : 0;
    const int rotb = (rows_are_ref == (oti_target == 0)) ? 0 : P.oti;
    const float *nrow = normtab + (role ? P.nr : P.nq) + (int64_t)rota * (MA + NGUARD);
    const float *ncol = normtab + (role ? P.nq : P.nr) + (int64_t)rotb * (MB + NGUARD);
    const float INF = __builtin_inff();

    // ---- MFMA operands straight from the rotated frame pool (rotpool_kernel): one 12-byte load per lane and 16-frame block
    const int lr = lane & 15, lk = lane >> 4;
    int c0a = lk - rota; if (c0a < 0) c0a += NBIN;
    int c0b = lk - rotb; if (c0b < 0) c0b += NBIN;
    constexpr unsigned PB = (unsigned)FROT * 4u;
    const unsigned offA = (unsigned)(((c0a >> 2) * NBIN + (c0a & 3) * 3) * 4);
    const unsigned offB = (unsigned)(((c0b >> 2) * NBIN + (c0b & 3) * 3) * 4);
    const char *pool_b = reinterpret_cast<const char *>(frot);
    const char *fra = pool_b + (role ? P.fr : P.fq) * (int64_t)PB + offA;
    const char *frb = pool_b + (role ? P.fq : P.fr) * (int64_t)PB + offB;
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    typedef f32x3 f32x3_u __attribute__((aligned(4)));
    float areg[3];
    {
        int f = i0 + lr;
        f = f > TA - 1 ? TA - 1 : f;           // rows beyond the matrix get a norm of +inf below
        const f32x3 v = *reinterpret_cast<const f32x3_u *>(fra + (size_t)f * PB);
        areg[0] = v.x; areg[1] = v.y; areg[2] = v.z;
    }
    float xrow[BAND];
#pragma unroll
    for (int a = 0; a < BAND; ++a) xrow[a] = nrow[i0 + a];      // (rows past the matrix: the table's +inf guard entries -> +inf cells)
    float *Sw = smem + wave * (16 * SP);
    const PctPos pp = role ? P.pos_q : P.pos_r;

    const int ntiles = (MB + BAND - 1 + 63) / 64;      // <= 4 NSTEP (8 / 12) by dispatch
    typedef float BvT[NCT][3];
    typedef f32x4 AccT[NCT];
#ifndef ACX_B2_BUFFER_LOADS
#define ACX_B2_BUFFER_LOADS 1      /* +0.5-1 % over plain global loads with their 64-bit VALU address arithmetic (profiles/r05_narrow_classes.md) */
#endif
#if ACX_B2_BUFFER_LOADS
    // the column-frame operands through a buffer descriptor: scalar tile offset + one 32-bit lane offset, no VALU address arithmetic per tile
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void *)(pool_b + ((role ? P.fq : P.fr) - 8) * (int64_t)PB), 0, -1, 0x00020000);
    const unsigned voffB = offB + (unsigned)lr * PB;
#endif
    auto load_operands = [&](int tile, BvT &bv, auto tb0_tag) {
        constexpr int tb0 = decltype(tb0_tag)::value;
#if ACX_B2_BUFFER_LOADS
        typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
        const unsigned so = (unsigned)(64 * tile - (BAND - 1) + 8) * PB;
#pragma unroll
        for (int tb = tb0; tb < NCT; ++tb) {
            const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rsB, voffB, so + 16u * PB * tb, 0);
            bv[tb][0] = __uint_as_float(v.x); bv[tb][1] = __uint_as_float(v.y); bv[tb][2] = __uint_as_float(v.z);
        }
#else
        const char *p = frb + (ptrdiff_t)(64 * tile - (BAND - 1) + lr) * (ptrdiff_t)PB;
#pragma unroll
        for (int tb = tb0; tb < NCT; ++tb) {
            const f32x3 v = *reinterpret_cast<const f32x3_u *>(p + 16 * PB * tb);
            bv[tb][0] = v.x; bv[tb][1] = v.y; bv[tb][2] = v.z;
        }
#endif
    };
    auto load_norms = [&](int tile, float (&yv)[BAND]) {
        typedef float f32x4n __attribute__((ext_vector_type(4), aligned(4)));
        const f32x4n *p = reinterpret_cast<const f32x4n *>(ncol + (64 * tile - (BAND - 1)) + lane);
        const f32x4n v0 = p[0], v1 = p[1];
        yv[0] = v0.x; yv[1] = v0.y; yv[2] = v0.z; yv[3] = v0.w;
        yv[4] = v1.x; yv[5] = v1.y; yv[6] = v1.z; yv[7] = v1.w;
    };
    auto gram = [&](const BvT &bv, AccT &acc, auto tb0_tag) {
        constexpr int tb0 = decltype(tb0_tag)::value;
#pragma unroll
        for (int tb = tb0; tb < NCT; ++tb) acc[tb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int tb = tb0; tb < NCT; ++tb)
                acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[tb][kb], areg[kb], acc[tb], 0, 0, 0);
    };
    auto walk = [&](const float (&sv)[M + BAND - 1], const float (&yv)[BAND], float (&dv)[BAND]) {
#pragma unroll
        for (int a = 0; a < BAND; ++a) {
            const float xy = tree_sum<M>(sv + a);
            float t3 = ROLE ? (__builtin_fmaf(-2.0f, xy, yv[a]) + xrow[a]) : (__builtin_fmaf(-2.0f, xy, xrow[a]) + yv[a]);
            if (!(t3 > 0.0f)) t3 = 0.0f;
            dv[a] = t3;
        }
    };
    const int pitchD = P.pitchD;
    float *D = scratch + P.offD + (size_t)i0 * pitchD;
    float xv[BAND][NSTEP];

    // ---- sweep: wave w takes tiles w cpw .. w cpw + cpw - 1 (cpw = ceil(ntiles / 4) <= 2); the other tile slots of the 8 that
    // make a row of 32 NV positions are padded with +inf (tile_of)
    constexpr int HB = NCT - 4;
    const int cpw = (ntiles + WAVES - 1) / WAVES;
    auto tile_of = [&](int st) { return st < cpw ? wave * cpw + st : WAVES * cpw + wave * (NSTEP - cpw) + (st - cpw); };
    BvT bv;
    f32x4 halo[HB];
    if (wave * cpw < ntiles) load_operands(wave * cpw, bv, std::integral_constant<int, 0>());
    static_for<0, NSTEP>([&](auto st_tag) {
        constexpr int st = decltype(st_tag)::value;
        constexpr int tb0 = st == 0 ? 0 : HB;
        const int tile = tile_of(st);
        if (st < cpw && tile < ntiles) {      // wave-uniform
            float yv[BAND];
            load_norms(tile, yv);
            if constexpr (st > 0) {
#pragma unroll
                for (int h = 0; h < HB; ++h) *reinterpret_cast<f32x4 *>(Sw + lr * SP + 16 * h + 4 * lk) = halo[h];
            }
            {
                AccT acc;
                gram(bv, acc, std::integral_constant<int, tb0>());
                if (st + 1 < NSTEP && st + 1 < cpw && tile + 1 < ntiles)
                    load_operands(tile + 1, bv, std::integral_constant<int, HB>());
#pragma unroll
                for (int tb = tb0; tb < NCT; ++tb) *reinterpret_cast<f32x4 *>(Sw + lr * SP + 16 * tb + 4 * lk) = acc[tb];
                if constexpr (st + 1 < NSTEP) {
#pragma unroll
                    for (int h = 0; h < HB; ++h) halo[h] = acc[4 + h];
                }
            }
            wave_lds_fence();
            float sv[M + BAND - 1], dv[BAND];
#pragma unroll
            for (int u = 0; u < M + BAND - 1; ++u) sv[u] = Sw[u * SP + lane + u];
            walk(sv, yv, dv);
#pragma unroll
            for (int a = 0; a < BAND; ++a) xv[a][st] = dv[a];
            if constexpr (write_d2) {
                const int j0 = 64 * tile - (BAND - 1) + lane;
#pragma unroll
                for (int a = 0; a < BAND; ++a) {
                    const int j = j0 + a;
                    if (i0 + a < MA && j >= 0 && j < pitchD) D[a * pitchD + j] = dv[a];
                }
            }
            wave_lds_fence();
        } else {
#pragma unroll
            for (int a = 0; a < BAND; ++a) xv[a][st] = INF;
        }
    });
    if constexpr (write_d2) {
        const int npad = pitchD - MB;
        for (int idx = tid; idx < BAND * npad; idx += 64 * WAVES) {
            const int a = idx / npad, j = MB + idx - a * npad;
            if (i0 + a < MA) D[a * pitchD + j] = INF;
        }
    }
#ifdef ACX_ABL
    { float keep_ = 0.0f;
      for (int a = 0; a < BAND; ++a) for (int st = 0; st < NSTEP; ++st) keep_ += xv[a][st];
      ACX_ABL_EXIT(1, keep_); }
#endif
    __syncthreads();     // all slabs dead -> the LDS becomes the 8 exchange rows
    // ---- exchange: band row a in POSITION order (position p = 64 tile + lane <-> column p - 7 + a); an owner lane's 16
    // positions are followed by 16 bytes of pad
    {
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            // position p = 64 tile + lane belongs to owner lane p / NV, slot p % NV
            int off;
            if constexpr (NV == 16) off = tile_of(st) * (4 * LNP) + lane + 4 * (lane >> 4);
            else { const int pp_ = 64 * tile_of(st) + lane, ow = pp_ / NV; off = pp_ + 4 * ow; }
            float *dst = smem + off;
#pragma unroll
            for (int a = 0; a < BAND; ++a) dst[a * ROWP] = xv[a][st];
        }
    }
    __syncthreads();
    // ---- wave w owns rows 2 w (lanes 0-31) and 2 w + 1 (lanes 32-63)
    const int l = lane & (GL - 1);
    const int grp = lane / GL;                         // which of the wave's RPW rows this lane works on
    const int myband = RPW * wave + grp;               // band row of this lane's group
    const int row = i0 + myband;
    float *myrow = smem + myband * ROWP;
    float xr[NV];
    {
        const float *mine = myrow + l * LNP;
#pragma unroll
        for (int j = 0; j < NV / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(mine + 4 * j);
            xr[4 * j + 0] = v.x; xr[4 * j + 1] = v.y; xr[4 * j + 2] = v.z; xr[4 * j + 3] = v.w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the rows have left LDS: from here on they are this wave's scratch
#ifdef ACX_ABL
    { float keep_ = 0.0f;
      for (int t = 0; t < NV; ++t) keep_ += xr[t];
      ACX_ABL_EXIT(2, keep_); }
#endif
    if (i0 + RPW * wave >= MA) return;                            // wave-uniform: none of the wave's rows exists
    const bool need = row < MA;                                   // (the last band of a matrix with an odd number of rows)
    const int cshift = (BAND - 1) - myband;                       // slot s of the row <-> column s - cshift
    const int n = MB;
    const float kf = pp.kf, fl = pp.fl, ce = pp.ce;
    const int ilo = pp.ilo, ihi = pp.ihi, k = pp.k;
    const bool interp = (pct_mode == 0 || pct_mode == 1);
    const bool want_next = interp && ihi != ilo;
    typedef __attribute__((address_space(3))) void lds_void;
    // a half's own row as scratch: 64 mailbox dwords, then the 256-bin histogram (B2Mail)
    const unsigned mb_addr = (unsigned)(uintptr_t)(lds_void *)myrow;
    const int end_valid = MB + cshift;                            // slots [cshift, end_valid) are cells
    const bool lane_has_data = l * NV < end_valid;
    const bool group_full = l * NV >= cshift && l * NV + NV <= end_valid;
    float slo = 0.0f, shi = 0.0f;
    unsigned okm = 0u;
#ifndef ACX_B2_FINE_BELOW
#define ACX_B2_FINE_BELOW 9       /* rows of fewer than 9 (rank + 2) cells -- under ~186 at kappa = 0.095 -- take the half-lane pivot (pair_select_pivot):
                                     i.i.d. tracks of 150 frames 115 -> 165 Gcells/s, of 100 frames 72 -> 100; from 200 frames on no difference at 9 / 12 / 16 */
#endif
    const bool fine_pivot = (ihi + 2) * ACX_B2_FINE_BELOW > n;
    const bool use_pivot = (ihi + 2) * (fine_pivot ? 5 : 9) <= n;
    if (use_pivot) {
        // zero the mailboxes + histogram: dwords [32, 64 + 256) of the row, 16 bytes per lane, three rounds of 32 lanes
        float *z = myrow + 32 + 4 * l;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < 2 || l < 8) *reinterpret_cast<float4 *>(z + 4 * GL * q) = make_float4(0.f, 0.f, 0.f, 0.f);      // [32, 64 + BINS)
        wave_lds_fence();
        const bool ok = pair_select_pivot<NV, GL>(xr, k, want_next, mb_addr, lane, slo, shi, lane_has_data, group_full, fine_pivot);
        const unsigned long long om = __ballot(ok);
#pragma unroll
        for (int g = 0; g < RPW; ++g) okm |= (unsigned)((om >> (GL * g)) & 1ull) << g;
    }
    // ---- rows the one-pass selection could not decide: the whole wave takes them one at a time through band_kernel's
    // fallbacks -- the row goes back to LDS and returns as 8 values per lane of all 64 lanes
    {
        const unsigned long long needm = __ballot(need);
        unsigned needb = 0u;
#pragma unroll
        for (int g = 0; g < RPW; ++g) needb |= (unsigned)((needm >> (GL * g)) & 1ull) << g;
        unsigned todo = needb & ~okm;
        if (todo != 0u) {
            // both rows of the wave are in registers: 2 ROWP floats of scratch; a 512-aligned block of 512 floats inside it
            // is the histogram of wave_select_fast, the rest holds the candidates and the generic selection's bins
            const int base = wave * RPW * ROWP;
            const int hoff = (base + 511) & ~511;
            float *hist = smem + hoff, *aux = smem + hoff + 512;                     // (hoff <= base + 256: both fit the two rows)
            float *relay = smem + base;                                              // the row in position order: 32 NV floats
            constexpr int NV8 = NV * GL / 64;                                        // values per lane of the whole wave
            static_assert(RPW * ROWP >= 256 + 512 + 512 && RPW * ROWP >= GL * NV, "fallback scratch fits the wave's rows");
            const unsigned fh_addr = (unsigned)(uintptr_t)(lds_void *)hist;
            for (int h = 0; h < RPW; ++h) {
                if (!((todo >> h) & 1u)) continue;
                const int cs = (BAND - 1) - (RPW * wave + h);
                wave_lds_fence();
                if (grp == h) {
#pragma unroll
                    for (int j = 0; j < NV / 4; ++j)
                        *reinterpret_cast<float4 *>(relay + l * NV + 4 * j) = make_float4(xr[4 * j], xr[4 * j + 1], xr[4 * j + 2], xr[4 * j + 3]);
                }
                wave_lds_fence();
                float x8[NV8];
#pragma unroll
                for (int j = 0; j < NV8 / 4; ++j) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(relay + NV8 * lane + 4 * j);
                    x8[4 * j] = v0.x; x8[4 * j + 1] = v0.y; x8[4 * j + 2] = v0.z; x8[4 * j + 3] = v0.w;
                }
                wave_lds_fence();
                for (int q = 0; q < 2; ++q) *reinterpret_cast<float4 *>(hist + 256 * q + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
                wave_lds_fence();
                float s_lo, s_hi;
                bool done = wave_select_fast<NV8, 512, 1>(x8, k, want_next, fh_addr, aux, lane, s_lo, s_hi, lane * NV8 < MB + cs);
                if (!done) {
                    unsigned *ghist = reinterpret_cast<unsigned *>(aux) + 64;
                    unsigned *counter = reinterpret_cast<unsigned *>(aux) + 64 + SelGeom<256>::SLOTS;
                    const SelectResult sr = wave_select_regs<NV8, 256>(x8, k, ghist, aux, counter, lane, interp);
                    s_lo = sr.value;
                    s_hi = (interp && ihi != ilo && sr.cnt_le <= ihi) ? sr.next : sr.value;
                }
                if (grp == h) { slo = s_lo; shi = s_hi; }
            }
            wave_lds_fence();
        }
    }
    ACX_ABL_EXIT(3, slo + shi);
    // ---- eps, the d2-domain threshold (band_row_tail's reasoning, per half) and, in the row pass, the bitmap
    float *X = thr + P.offX;
    float thr_row;
    const bool weights_ok = interp && ihi == ilo + 1 && inclusive && fl >= 1.0f && (ce - kf) >= 0.00390625f && (kf - fl) >= 0.00390625f;
    const bool easy = !want_eps && weights_ok && shi < INF && (shi - slo) > shi * 0.000244140625f;
    if (__ballot(easy || !need) == ~0ull) {
        thr_row = slo;
        if (l == 0 && need) X[role ? P.pitchT + row : row] = slo;
    } else {
        // (rare: ties, an exact-integer position, the debug entry point -- every lane works on its own half's values)
        const float dlo = __builtin_sqrtf(slo), dhi = __builtin_sqrtf(shi);
        float eps = dlo;
        if (interp && !(pct_mode == 0 && ihi == ilo)) {
            const float d0 = __fmul_rn(dlo, __fsub_rn(ce, kf));
            const float d1 = __fmul_rn(dhi, __fsub_rn(kf, fl));
            eps = __fadd_rn(d0, d1);
        }
        if (inclusive && dlo <= eps && eps < dhi) thr_row = slo;
        else thr_row = d2_threshold(eps, inclusive);
        if (l == 0 && need) {
            const int o = role ? P.pitchT + row : row;
            X[o] = thr_row;
            X[P.pitchT + P.pitchD + o] = eps;
        }
    }
    ACX_ABL_EXIT(4, thr_row);
    if constexpr (ROLE == 0) {
        if (bits && need) {
            typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
            const float *tc = X + P.pitchT + (l * NV - cshift);
            float tcv[NV];
#pragma unroll
            for (int j = 0; j < NV / 4; ++j) {
                const f32x4_u v = *reinterpret_cast<const f32x4_u *>(tc + 4 * j);
                tcv[4 * j + 0] = v.x; tcv[4 * j + 1] = v.y; tcv[4 * j + 2] = v.z; tcv[4 * j + 3] = v.w;
            }
            int lo = cshift - l * NV, hi = MB + cshift - l * NV;
            lo = lo < 0 ? 0 : lo;
            hi = hi > NV ? NV : hi;
            unsigned valid = 0u;
            if (hi > lo) valid = ((1u << (hi - lo)) - 1u) << lo;      // (hi - lo <= NV < 32)
            unsigned acc = 0u;
#pragma unroll
            for (int t = NV - 1; t >= 0; --t) {
                float mthr;
                asm("v_min_f32 %0, %1, %2" : "=v"(mthr) : "v"(tcv[t]), "v"(thr_row));
                asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(xr[t]), "v"(mthr) : "vcc");
            }
            acc &= valid;
            unsigned *rowbits = reinterpret_cast<unsigned *>(bits + P.offT + (size_t)row * P.nw);
            const int ndw = 2 * P.nw;
            if constexpr (NV == 16) {
                acc |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)acc, 0xF5, 0xf, 0xf, false) << 16;    // quad_perm [1,1,3,3]: the odd neighbour's 16 bits
                const int d = l >> 1;
                if ((l & 1) == 0 && d < ndw) rowbits[d] = acc;
            } else {
                // 24 bits per lane: a quad of lanes makes three dwords -- lane j of the quad stores dword j = its own bits from 8 j on,
                // topped up with the next lane's (quad_perm [1,2,3,3]); lane 3 stores nothing
                const unsigned nxt = (unsigned)__builtin_amdgcn_update_dpp(0, (int)acc, 0xF9, 0xf, 0xf, false);
                const int j = l & 3;
                const unsigned dwv = (acc >> (8 * j)) | (nxt << (24 - 8 * j));
                const int d = 3 * (l >> 2) + j;
                if (j < 3 && d < ndw) rowbits[d] = dwv;
            }
            for (int z = NV * GL / 32 + l; z < ndw; z += GL) rowbits[z] = 0u;     // (words beyond this size class: none by dispatch)
        }
    }
}

}  // namespace acx
