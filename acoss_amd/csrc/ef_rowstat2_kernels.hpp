// EarlyFusion row statistics of the three cross-similarity matrices, TWO matrix rows per wave (round 5).
//
// Reference: csm_to_binary's row threshold (cross_recurrence.py:136-161) and getWCSM's row neighbourhood mean
// (similarity_fusion.py:38-46) of every row of C -- ef_rowstat_kernel<2, false> (ef_kernels.hpp) does them with one wave per
// row of <= 512 cells, 390 VALU instructions per row of which the selection's reductions, scans, bin search and ranking do not
// depend on how many values a lane holds.  Here, as in band2_kernel (serra09_band2_kernels.hpp): lanes 0-31 hold one row, lanes
// 32-63 the next, 16 values per lane; the two halves talk through LDS mailboxes; both order statistics of a row (rank K - 1 for
// the neighbourhood mean, rank kappa N - 1 for the threshold) come out of ONE pivot-filtered histogram as the smallest / largest
// cell of their bin (three or more cells around the rank: gathered and ranked inside the half).  Same values as the one-row
// kernel's selection, and the neighbourhood mean adds its cells in the one-row kernels' order (two accumulators per lane, see
// the kernel): bit-identical statistics whichever kernel a batch takes.  A row this pass cannot decide (surplus ties at the
// threshold, too few cells below the pivot, a short row) is re-laid out through LDS and goes through ef_row_finish, the one-row
// kernel's own code.
#pragma once
#include "ef_kernels.hpp"
#include "serra09_band2_kernels.hpp"

namespace acx {

struct EfMail {            // dword offsets inside a half's mailbox area (zeroed before use)
    static constexpr int CAND = 0;      // 32 candidate slots of the crowded-bin path
    static constexpr int COUNTER = 32;  // their counter
    static constexpr int E = 36;        // [36] max pattern of bin1's cells, [37] max complemented pattern (-> its minimum), [38], [39]: bin2
    static constexpr int OWN = 40;      // [40, 41] {L1 + 1, excl}, [42, 43] {L2 + 1, excl}
    static constexpr int BIN = 44;      // [44..46] {bin1, below1, count1}, [48..50] {bin2, below2, count2}
    static constexpr int RES = 52;      // result of the crowded-bin path
    static constexpr int HIST = 64;     // 256 bins
    static constexpr int DWORDS = 64 + 256;
};

__device__ __forceinline__ float half_sum_f(float s)      // the same in every lane of a half: lanes 1, 2 apart, mirrored 8 / 16, then the two 16-lane rows
{
    s += ef_dpp_f<EF_DPP_XOR1>(s);
    s += ef_dpp_f<EF_DPP_XOR2>(s);
    s += ef_dpp_f<EF_DPP_HALF_MIRROR>(s);
    s += ef_dpp_f<EF_DPP_ROW_MIRROR>(s);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int half_sum_i(int v) { return half_allreduce(v, 0, OpAddI()); }

// The members of ONE bin of a half (pattern `a`; 0: this half has nothing to gather), at most 32, through an LDS counter into the
// half's candidate slots, ranked directly; the value of rank `want` among them in every lane of the half.
__device__ __forceinline__ float half_gather_rank(const float (&x)[B2_NV], float scale, float offm, unsigned a, int nc, int want,
                                                  unsigned mb_addr, int l)
{
    const float INF = __builtin_inff();
    const unsigned cand_addr = mb_addr + 4u * EfMail::CAND;
    lds_u32 *counter = (lds_u32 *)(mb_addr + 4u * EfMail::COUNTER);
#pragma unroll
    for (int t = 0; t < B2_NV; ++t) {
        if (__float_as_uint(__builtin_fmaf(x[t], scale, offm)) == a) {
            const unsigned pos = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            *(lds_f32 *)(cand_addr + 4u * (pos & 31u)) = x[t];
        }
    }
    wave_lds_fence();
    if (l >= nc) *(lds_f32 *)(cand_addr + 4u * (unsigned)l) = INF;
    if (l == 0) *counter = 0u;                          // (ready for the other rank)
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
    if (l < nc && rank == want) *(lds_f32 *)(mb_addr + 4u * EfMail::RES) = mine;
    wave_lds_fence();
    const float r = *(const lds_f32 *)(mb_addr + 4u * EfMail::RES);
    wave_lds_fence();
    return r;
}

// Ranks k1 <= k2 (0-based) of the two rows a wave holds (pair_select_pivot of band2, two INDEPENDENT ranks).  `mb_addr`: the lane's
// own half's zeroed EfMail area.  Returns per lane whether its half has both values.
__device__ __forceinline__ bool ef_pair_select2(const float (&x)[B2_NV], int k1, int k2, unsigned mb_addr, int lane, float &v1, float &v2,
                                                bool lane_has_data, bool group_full, float delta)
{
    constexpr int NV = B2_NV, NB = 256, BPL = 8;
    const float INF = __builtin_inff();
    const int l = lane & 31;
    const unsigned hist_addr = mb_addr + 4u * EfMail::HIST;
    unsigned mnl = 0xFFFFFFFFu;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const unsigned b = __float_as_uint(x[t]);
        mnl = b < mnl ? b : mnl;
    }
    const unsigned mnu = (unsigned)half_allreduce((int)mnl, -1, OpMinU());
    const int mxg = half_allreduce(group_full ? (int)mnl : (int)0x80000000, (int)0x80000000, OpMaxI());
    const float mn = __uint_as_float(mnu);
    const float gm = __uint_as_float((unsigned)mxg);
    const float range = __builtin_fmaf(delta, gm - mn, gm) - mn;
    const bool good = mxg >= 0 && (int)mnu >= 0 && range >= 1e-30f && range <= 1e30f && mn <= 2048.0f * range;
    constexpr unsigned MAGIC = 0x4B000000u;
    const float scale = ((float)NB - 3.0f) * __builtin_amdgcn_rcpf(range);
    const float offm = (lane_has_data && good) ? (8388609.0f - mn * scale) : INF;
#ifndef ACX_EF2_KEEP_PAT
#define ACX_EF2_KEEP_PAT 1
#endif
#if ACX_EF2_KEEP_PAT
    unsigned patk[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) patk[t] = __float_as_uint(__builtin_fmaf(x[t], scale, offm));
#define ACX_EF2_PAT(t_) patk[t_]
#else
#define ACX_EF2_PAT(t_) __float_as_uint(__builtin_fmaf(x[t_], scale, offm))
#endif
    {
        const unsigned nb = __builtin_amdgcn_readfirstlane(MAGIC + NB);
        const unsigned hb = hist_addr - 4u * MAGIC;
        unsigned one = 1u;
        asm volatile("" : "+v"(one));
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            unsigned long long m0, m1, m2, m3, sv;
            unsigned a0, a1, a2, a3;
            unsigned pat[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) pat[u] = ACX_EF2_PAT(t + u);
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
                         : [q0] "v"(pat[0]), [q1] "v"(pat[1]), [q2] "v"(pat[2]), [q3] "v"(pat[3]),
                           [nb] "s"(nb), [hb] "v"(hb), [one] "v"(one)
                         : "memory");
        }
    }
    wave_lds_fence();
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
    const int incl = half_incl_scan_i(lsum);
    const int excl = incl - lsum;
    if (excl <= k1 && k1 < incl) *(lds_u32x2 *)(mb_addr + 4u * EfMail::OWN) = u32x2{(unsigned)l + 1u, (unsigned)excl};
    if (excl <= k2 && k2 < incl) *(lds_u32x2 *)(mb_addr + 4u * (EfMail::OWN + 2)) = u32x2{(unsigned)l + 1u, (unsigned)excl};
    wave_lds_fence();
    const int e = l & 15;
    const bool second = (l & 16) != 0;
    const u32x2 own = *(const lds_u32x2 *)(mb_addr + 4u * EfMail::OWN + (second ? 8u : 0u));
    const int Lx = (int)own.x - 1, exx = (int)own.y;
    const int kk = second ? k2 : k1;
    int c = (int)*(const lds_u32 *)(hist_addr + (unsigned)(Lx * (BPL * 4) + (e & 7) * 4));
    c = e < BPL ? c : 0;
    int P = c;
    P += __builtin_amdgcn_update_dpp(0, P, 0x111, 0xf, 0xf, false);
    P += __builtin_amdgcn_update_dpp(0, P, 0x112, 0xf, 0xf, false);
    P += __builtin_amdgcn_update_dpp(0, P, 0x114, 0xf, 0xf, false);
    {
        const int below = exx + P - c;
        if (e < BPL && Lx >= 0 && below <= kk && kk < below + c) {
            typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
            typedef __attribute__((address_space(3))) u32x3 lds_u32x3;
            *(lds_u32x3 *)(mb_addr + 4u * EfMail::BIN + (second ? 16u : 0u)) = u32x3{(unsigned)(Lx * BPL + e), (unsigned)below, (unsigned)c};
        }
    }
    wave_lds_fence();
    const u32x4 r1 = *(const lds_u32x4 *)(mb_addr + 4u * EfMail::BIN), r2 = *(const lds_u32x4 *)(mb_addr + 4u * (EfMail::BIN + 4));
    const int bin1 = (int)r1.x, cnt1 = (int)r1.z, bin2 = (int)r2.x, cnt2 = (int)r2.z;
    const int p1 = k1 - (int)r1.y, p2 = k2 - (int)r2.y;              // position of the rank inside its bin
    const bool found = good && cnt1 != 0 && cnt2 != 0;
    const unsigned a1 = found ? MAGIC + (unsigned)bin1 : 0u, a2 = found ? MAGIC + (unsigned)bin2 : 0u;
    {
        lds_u32 *ext = (lds_u32 *)(mb_addr + 4u * EfMail::E);
#pragma unroll
        for (int t = 0; t < NV; t += 4) {
            bool h1[4], h2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned pt = ACX_EF2_PAT(t + u);
                h1[u] = pt == a1; h2[u] = pt == a2;
            }
            if (__ballot(h1[0] || h1[1] || h1[2] || h1[3] || h2[0] || h2[1] || h2[2] || h2[3]) != 0ull) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned xb = __float_as_uint(x[t + u]);
                    if (h1[u]) {
                        __hip_atomic_fetch_max(ext + 0, xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_max(ext + 1, ~xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (h2[u]) {
                        __hip_atomic_fetch_max(ext + 2, xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_max(ext + 3, ~xb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
    }
    wave_lds_fence();
    const u32x4 ev = *(const lds_u32x4 *)(mb_addr + 4u * EfMail::E);
    bool ok1 = found && (p1 == 0 || p1 == cnt1 - 1), ok2 = found && (p2 == 0 || p2 == cnt2 - 1);
    v1 = __uint_as_float(p1 == 0 ? ~ev.y : ev.x);
    v2 = __uint_as_float(p2 == 0 ? ~ev.w : ev.z);
    // three or more cells in a bin with the rank in their middle: gather that bin's members inside the half
    const bool crowd1 = found && !ok1 && cnt1 <= 32, crowd2 = found && !ok2 && cnt2 <= 32;
    if (__ballot(crowd1) != 0ull) {
        const float r = half_gather_rank(x, scale, offm, crowd1 ? a1 : 0u, crowd1 ? cnt1 : 0, p1, mb_addr, l);
        if (crowd1) { v1 = r; ok1 = true; }
    }
    if (__ballot(crowd2) != 0ull) {
        const float r = half_gather_rank(x, scale, offm, crowd2 ? a2 : 0u, crowd2 ? cnt2 : 0, p2, mb_addr, l);
        if (crowd2) { v2 = r; ok2 = true; }
    }
    wave_lds_fence();
    return ok1 && ok2;
}

// The cold path of ef_rowstat2_kernel: one row, already in column order in `relay` (512 floats of LDS), through ef_row_finish.
#ifndef ACX_EF2_FALLBACK_INLINE
#define ACX_EF2_FALLBACK_INLINE 1
#endif
#if ACX_EF2_FALLBACK_INLINE
__device__ __forceinline__
#else
__device__ __attribute__((noinline))
#endif
void ef_rowstat2_fallback(const float *relay, const EfPair *pp, int s, int kw, int row, float *__restrict__ stat,
                                                               unsigned *__restrict__ bits, unsigned *fhist_w, unsigned *hist_w, float *cand_w,
                                                               unsigned *counter_w, int lane)
{
    const EfPair P = *pp;
    float x8[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float4 v4 = *reinterpret_cast<const float4 *>(relay + 256 * q + 4 * lane);
        x8[4 * q + 0] = v4.x; x8[4 * q + 1] = v4.y; x8[4 * q + 2] = v4.z; x8[4 * q + 3] = v4.w;
    }
    wave_lds_fence();
    ef_row_finish<2, false>(x8, P, s, 0, kw, P.N, P.pitchC, row, stat, bits, fhist_w, hist_w, cand_w, counter_w, lane);
}

// mode 0 of ef_rowstat_kernel<2, false> (rows of C: threshold + tie column, neighbourhood mean, binarised row) for matrices whose
// rows hold <= 512 cells: grid = (ceil(M / 8), pairs, features), 256 threads, 8 rows per workgroup.
__global__ __launch_bounds__(256, 6) void ef_rowstat2_kernel(const EfPair *__restrict__ pd, float *__restrict__ scratch, float *__restrict__ stat,
                                                            unsigned *__restrict__ bits, int kw)
{
    constexpr int NV = B2_NV;
    __shared__ __attribute__((aligned(4096))) unsigned fhist[4][256];                       // the fallback's scratch (ef_row_finish), per wave
    __shared__ __attribute__((aligned(16))) unsigned hist[4][SelGeom<EF_ROW_GB>::SLOTS];    // (also: the re-layout buffer, 512 floats)
    __shared__ __attribute__((aligned(16))) float cand[4][64];
    __shared__ unsigned counter[4];
    __shared__ __attribute__((aligned(16))) unsigned mail[8][EfMail::DWORDS];               // per half
    static_assert(SelGeom<EF_ROW_GB>::SLOTS >= 512, "the re-layout buffer");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const EfPair P = pd[blockIdx.y];
    const int s = blockIdx.z;
    const int n = P.N, pitch = P.pitchC;
    const int row0 = blockIdx.x * 8 + 2 * wave;
    if (row0 >= P.M) return;
    const int l = lane & 31, hh = lane >> 5;
    const int row = row0 + hh;
    const bool need = row < P.M;
    const float INF = __builtin_inff();
    float x[NV];                                   // x[4 q + e] = column 128 q + 4 l + e of the half's row
    {
        const float *v = scratch + ef_c_off(P, s) + (size_t)(need ? row : row0) * pitch;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = 128 * q + 4 * l;
            float4 t = make_float4(INF, INF, INF, INF);
            if (j < pitch) t = *reinterpret_cast<const float4 *>(v + j);
            x[4 * q + 0] = (j + 0 < n) ? t.x : INF;
            x[4 * q + 1] = (j + 1 < n) ? t.y : INF;
            x[4 * q + 2] = (j + 2 < n) ? t.z : INF;
            x[4 * q + 3] = (j + 3 < n) ? t.w : INF;
        }
    }
    // a lane takes part in the pivot estimate when at least three quarters of the slots it CAN have in a row of n are cells
    int cells = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = 128 * q + 4 * l;
        cells += j + 3 < n ? 4 : (j < n ? n - j : 0);
    }
    const bool lane_has_data = cells > 0;
    const int cap = 4 * ((n + 127) / 128);
    const bool group_full = cells >= cap - cap / 4;
    const int kk = kw < n ? kw : n, kb = P.kbin;
    const bool two = kb > kk && kb < n && (kb + 1) * 6 <= n;         // (the condition under which ef_rowstat_kernel takes both ranks from one histogram)
    typedef __attribute__((address_space(3))) void lds_void;
    unsigned *mymail = mail[2 * wave + hh];
    const unsigned mb_addr = (unsigned)(uintptr_t)(lds_void *)mymail;
    bool ok = false;
    float vk = 0.0f, t = 0.0f;
    if (two) {
#pragma unroll
        for (int q = 0; q < 3; ++q)                                  // 320 dwords: 16 bytes per lane, 32 + 32 + 16 lanes
            if (q < 2 || l < 16) *reinterpret_cast<uint4 *>(mymail + 128 * q + 4 * l) = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_fence();
        ok = ef_pair_select2(x, kk - 1, kb - 1, mb_addr, lane, vk, t, lane_has_data, group_full, 0.15f);
    }
    // neighbourhood mean: the cells below the kk-th smallest + (kk - their number) times that value (mean_k_smallest)
    // -- in the ORDER of the one-row kernels (mean_k_smallest + ef_wave_sum_f, rows of 8 or 16 values per lane of 64): a pair's
    // score must not depend on which statistics kernel its batch takes (a batch with one row of more than 512 cells runs
    // ef_rowstat_kernel<4> for ALL its pairs).  One-row lane l holds columns 4 l + e and 256 + 4 l + e -- this half's groups
    // q = 0 and 2 -- and lane 32 + l the groups q = 1 and 3; the wave's tree is four steps inside every 16-lane row, then
    // (r0 + r1) + (r2 + r3): the same four steps on both accumulators, then (A_r0 + A_r1) + (B_r0 + B_r1).
    float m;
    {
        float accA = 0.0f, accB = 0.0f;
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool lt = x[4 * q + e] < vk;
                if (q & 1) accB += lt ? x[4 * q + e] : 0.0f;
                else accA += lt ? x[4 * q + e] : 0.0f;
                cnt += lt ? 1 : 0;
            }
        const int tot = half_sum_i(cnt);
        const float ssum = half_sum_f(accA) + half_sum_f(accB);
        m = (ssum + (float)(kk - tot) * vk) / (float)kk;
    }
    // cells at or below the threshold: exactly kb unless ties straddle it (then the row takes the one-row code)
    {
        int le = 0;
#pragma unroll
        for (int e = 0; e < NV; ++e) le += x[e] <= t ? 1 : 0;
        ok = ok && half_sum_i(le) <= kb && t < INF;
    }
    if (ok && need) {
        float *S = stat + P.offS + s * ef_s_stride(P);
        if (l == 0) {
            S[P.pitchT + row] = m;
            S[row] = t;
            reinterpret_cast<int *>(S)[ef_jcut_off(P, s) + row] = 0x7fffffff;
        }
        // the binarised row: bit j % 32 of word j / 32; a lane's nibble of every 128-column group, eight lanes to a word
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned nib = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) nib |= x[4 * q + e] <= t ? (1u << e) : 0u;
            unsigned w = nib << (4 * (l & 7));
            w |= (unsigned)ef_dpp_i<EF_DPP_XOR1>((int)w);
            w |= (unsigned)ef_dpp_i<EF_DPP_XOR2>((int)w);
            w |= (unsigned)ef_dpp_i<EF_DPP_HALF_MIRROR>((int)w);
            const int word = 4 * q + (l >> 3);
            if ((l & 7) == 0 && 32 * word < pitch)
                bits[P.offB + ((int64_t)s * P.M + row) * (pitch >> 5) + word] = w;
        }
    }
    // ---- rows this pass could not decide: one at a time through the one-row code, 8 values per lane of all 64 lanes
    // (a real call: inlined, the cold path's registers -- a second copy of the row, the one-row selection's patterns -- cost
    // the hot path 15 spilled registers)
    const unsigned long long okm = __ballot(ok || !need);
    const unsigned todo = (((unsigned)okm & 1u) ? 0u : 1u) | (((unsigned)(okm >> 32) & 1u) ? 0u : 2u);
    if (todo != 0u) {
        float *relay = reinterpret_cast<float *>(hist[wave]);
        for (int f = 0; f < 2; ++f) {
            if (!((todo >> f) & 1u)) continue;
            wave_lds_fence();
            if (hh == f) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(relay + 128 * q + 4 * l) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
            }
            wave_lds_fence();
            ef_rowstat2_fallback(relay, pd + blockIdx.y, s, kw, row0 + f, stat, bits, fhist[wave], hist[wave], cand[wave], &counter[wave], lane);
        }
    }
}

}  // namespace acx
