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
//                         per row (one histogram pass), recurrence bitmap; D2 never reaches HBM
//   K3b qmax_bits_kernel  Qmax / Dmax row sweep over the recurrence bitmap, one wave per pair
// A/B pipeline kept behind ACX_PIPELINE=v1 (materialised D2 + D2^T, 20 B/cell):
//   K1 csm_tile_kernel, K2 rowsel_kernel, K3 qmax_kernel
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
constexpr int TILE = 64;       // output tile edge of K1
constexpr int STILE = 80;      // S tile edge (TILE + 16 halo): supports m <= 17
constexpr int SPITCH = 81;
#ifndef ACX_SEL_BINS
#define ACX_SEL_BINS 2048
#endif
constexpr int SEL_BINS = ACX_SEL_BINS; // histogram bins of the percentile selection
constexpr int MAX_M = 16;

struct PairDesc {
    int32_t q, r;          // track indices (query, reference)
    int32_t Tq, Tr;        // pooled lengths
    int32_t Mq, Mr;        // embedded lengths (rows, cols of the matrix)
    int32_t oti;           // filled by K0
    int32_t pitchD;        // row pitch of D2  (floats, multiple of 64, >= Mr)
    int32_t pitchT;        // row pitch of D2T (floats, multiple of 64, >= Mq)
    int32_t nw;            // 64-bit words per row of the recurrence bitmap (band pipeline)
    int64_t offD, offT;    // offD: float offset of D2 in the scratch arena (v1 / debug);
                           // offT: v1: float offset of D2^T; band pipeline: u64-word offset of the
                           // pair's recurrence bitmap (Mq rows x nw words) in the bit arena
    int64_t offX;          // float offset into the threshold arena:
                           //   [thr rows: pitchT][thr cols: pitchD][eps rows: pitchT][eps cols: pitchD]
};

__device__ __forceinline__ float wave_shfl(float v, int src)
{
    return __shfl(v, src, 64);
}

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
static __global__ void oti_kernel(PairDesc *pd, int B, const float *__restrict__ gch, int oti_on, int oti_target)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= B) return;
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

// ------------------------------------------------------------------------------------
// K1: squared embedded distances, 64x64 tile per workgroup (256 threads = 4 waves).
// ------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M>
__global__ __launch_bounds__(256) void csm_tile_kernel(const float *__restrict__ pool,
                                                       const int64_t *__restrict__ toff,
                                                       const PairDesc *__restrict__ pd,
                                                       float *__restrict__ scratch,
                                                       int tiles_x, int oti_target)
{
    __shared__ float Qs[NBIN][STILE];   // chroma tiles, bin-major (conflict-free MFMA operand reads)
    __shared__ float Rs[NBIN][STILE];
    __shared__ float NQ[STILE], NR[STILE];
    __shared__ float XX[TILE], YY[TILE];
    __shared__ float S[STILE * SPITCH];  // frame Gram tile; reused as the transpose stage

    const PairDesc P = pd[blockIdx.y];
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int i0 = ty * TILE, j0 = tx * TILE;
    if (i0 >= P.Mq || j0 >= P.Mr) return;   // block-uniform

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *qf = pool + toff[P.q] * NBIN;
    const float *rf = pool + toff[P.r] * NBIN;
    const int rotq = (oti_target == 1) ? P.oti : 0;
    const int rotr = (oti_target == 0) ? P.oti : 0;

    // ---- stage chroma (rotation = roll right: dst bin (c + s) % 12 <- src bin c)
    for (int idx = tid; idx < STILE * NBIN; idx += 256) {
        int a = idx / NBIN, c = idx - a * NBIN;
        int fq = i0 + a, fr = j0 + a;
        float vq = (fq < P.Tq) ? qf[(size_t)fq * NBIN + c] : 0.0f;
        float vr = (fr < P.Tr) ? rf[(size_t)fr * NBIN + c] : 0.0f;
        int cq = c + rotq; if (cq >= NBIN) cq -= NBIN;
        int cr = c + rotr; if (cr >= NBIN) cr -= NBIN;
        Qs[cq][a] = vq;
        Rs[cr][a] = vr;
    }
    __syncthreads();

    // ---- frame norms (fmaf chain over the bins, rotated order)
    if (tid < 2 * STILE) {
        const bool isq = tid < STILE;
        const int a = isq ? tid : tid - STILE;
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < NBIN; ++c) {
            float v = isq ? Qs[c][a] : Rs[c][a];
            acc = fmaf(v, v, acc);
        }
        if (isq) NQ[a] = acc; else NR[a] = acc;
    }

    // ---- frame Gram on the matrix cores: 5x5 tiles of 16x16, K = 12 in 3 k-steps
    {
        const int lr = lane & 15, lk = lane >> 4;
        for (int t = wave; t < 25; t += 4) {
            const int ta = t / 5, tb = t - ta * 5;
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
                float av = Qs[4 * kb + lk][16 * ta + lr];
                float bv = Rs[4 * kb + lk][16 * tb + lr];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                S[(16 * ta + 4 * lk + reg) * SPITCH + 16 * tb + lr] = acc[reg];
        }
    }
    __syncthreads();

    // ---- embedded norms of the tile's rows / columns
    if (tid < 2 * TILE) {
        const bool isq = tid < TILE;
        const int a = isq ? tid : tid - TILE;
        float s[M];
#pragma unroll
        for (int k = 0; k < M; ++k) s[k] = isq ? NQ[a + k] : NR[a + k];
        float v = tree_sum<M>(s);
        if (isq) XX[a] = v; else YY[a] = v;
    }
    __syncthreads();

    // ---- window sums along the diagonal + distance, 16 rows per thread
    const int c = lane;          // tile column
    const int rq = wave;         // 16-row quarter
    float o[16];
    {
        const float yy = YY[c];
        const bool colok = (j0 + c) < P.Mr;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int a = 16 * rq + t;
            float s[M];
#pragma unroll
            for (int k = 0; k < M; ++k) s[k] = S[(a + k) * SPITCH + c + k];
            const float xy = tree_sum<M>(s);
            const float t1 = 2.0f * xy;
            const float t2 = XX[a] - t1;
            float t3 = t2 + yy;
            if (!(t3 > 0.0f)) t3 = 0.0f;
            o[t] = colok ? t3 : __builtin_inff();
        }
    }
    // row-major store (256 B contiguous per wave-instruction)
    {
        float *D = scratch + P.offD;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = i0 + 16 * rq + t;
            if (i < P.Mq) D[(size_t)i * P.pitchD + j0 + c] = o[t];
        }
    }
    __syncthreads();   // all S reads done -> reuse as transpose stage O[64][65]
#pragma unroll
    for (int t = 0; t < 16; ++t) S[(16 * rq + t) * 65 + c] = o[t];
    __syncthreads();
    {
        float *DT = scratch + P.offT;
        const bool iok = (i0 + c) < P.Mq;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int b = 16 * rq + t;      // tile column -> row of D2T
            const int j = j0 + b;
            float v = S[c * 65 + b];
            if (j < P.Mr) DT[(size_t)j * P.pitchT + i0 + c] = iok ? v : __builtin_inff();
        }
    }
}

// ------------------------------------------------------------------------------------
// K2: per-row kappa-percentile threshold.  One wave per row; 4 waves per workgroup,
// each with a private LDS histogram.  No __syncthreads (waves are independent); LDS
// operations of one wave execute in order.
// ------------------------------------------------------------------------------------
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
#if defined(ACX_SEL_STOP) && ACX_SEL_STOP == 1
    return SelectResult{mn + mx, 0, INF};
#endif
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
#if defined(ACX_SEL_STOP) && ACX_SEL_STOP == 2
        return SelectResult{(float)hist[lane] + (float)bins[3], 0, INF};
#endif
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
#if defined(ACX_SEL_STOP) && ACX_SEL_STOP == 3
        return SelectResult{(float)(binsel + cum + cnt) + (float)bins[3], 0, INF};
#endif
        if (cnt <= 64) {
            // append the bin's elements (<= 64) to cand[] (order irrelevant), rank, pick.  Only
            // a handful of the NV register slots hold a hit in any lane: a wave-uniform ballot
            // skips the others.
#if !(defined(ACX_SEL_STOP) && ACX_SEL_STOP == 41)
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
#endif
            wave_lds_fence();
            const float mine = (lane < cnt) ? cand[lane] : INF;
            int rank = 0;
#pragma unroll 1
            for (int t = 0; t < cnt; ++t) {      // cnt is an SGPR: scalar loop, no LDS traffic
#if defined(ACX_SEL_STOP) && ACX_SEL_STOP == 42
                break;
#endif
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
#if defined(ACX_SEL_STOP) && ACX_SEL_STOP == 4
    return res;
#endif
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

// row in HBM -> registers (V4 back-to-back 16-byte loads per lane), then select
template <int V4>
__device__ __forceinline__ SelectResult wave_select(const float *__restrict__ v, int n, int pitch, int k,
                                                    unsigned *hist, float *cand, unsigned *counter,
                                                    int lane, bool want_next)
{
    constexpr int NV = 4 * V4;
    float x[NV];
    const float INF = __builtin_inff();
#pragma unroll
    for (int q = 0; q < V4; ++q) {
        const int j = 256 * q + 4 * lane;
        float4 t = make_float4(INF, INF, INF, INF);
        if (j < pitch) t = *reinterpret_cast<const float4 *>(v + j);
        x[4 * q + 0] = (j + 0 < n) ? t.x : INF;
        x[4 * q + 1] = (j + 1 < n) ? t.y : INF;
        x[4 * q + 2] = (j + 2 < n) ? t.z : INF;
        x[4 * q + 3] = (j + 3 < n) ? t.w : INF;
    }
    return wave_select_regs<NV>(x, k, hist, cand, counter, lane, want_next);
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

#ifdef ACX_TIMING
#define ACX_TS(k) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (tsel) tsel[k] = __builtin_readcyclecounter(); } while (0)
#else
#define ACX_TS(k) do { } while (0)
#endif
// NEGPAD: pads are negative instead of +inf.  They are skipped by the range pass without the bias
// add, fall into bin 0 (the conversion saturates at 0) and rank below every cell, so the caller
// passes k already raised by the number of pads that take part in the histogram.
template <int NV, int BINS, int COPIES = 1, bool NEGPAD = false>
__device__ __forceinline__ bool wave_select_fast(const float (&x)[NV], int k, bool want_next, unsigned hist_addr,
                                                 float *cand, int lane, float &slo, float &shi,
                                                 unsigned long long *tsel = nullptr, bool lane_has_data = true)
{
    ACX_TS(0);
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
    ACX_TS(1);
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
    ACX_TS(2);
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
    ACX_TS(3);
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
    ACX_TS(4);
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
    ACX_TS(5);
    return true;
}

// eps from the two order statistics d2_(ilo) <= d2_(ihi) (oracle percentile_f32)
__device__ __forceinline__ float percentile_eps2(float slo, float shi, int pct_mode, int ilo, int ihi,
                                                 float kf, float fl, float ce)
{
    if (!(pct_mode == 0 || pct_mode == 1)) return __builtin_sqrtf(slo);
    const float dlo = __builtin_sqrtf(slo), dhi = __builtin_sqrtf(shi);
    if (pct_mode == 0 && ihi == ilo) return dlo;
    const float d0 = __fmul_rn(dlo, __fsub_rn(ce, kf));
    const float d1 = __fmul_rn(dhi, __fsub_rn(kf, fl));
    return __fadd_rn(d0, d1);
}

template <int V4>
__global__ __launch_bounds__(256) void rowsel_kernel(const PairDesc *__restrict__ pd,
                                                     const float *__restrict__ scratch,
                                                     float *__restrict__ thr,
                                                     float kappa, int pct_mode, int inclusive)
{
    __shared__ __attribute__((aligned(16))) unsigned hist[4][SEL_SLOTS];
    __shared__ float cand[4][64];
    __shared__ unsigned counter[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const PairDesc P = pd[blockIdx.y];
    const int r = blockIdx.x * 4 + wave;
    if (r >= P.Mq + P.Mr) return;   // wave-uniform; no workgroup barriers below
    const bool side = r >= P.Mq;    // false: row of D2 (query frame), true: row of D2T (reference frame)
    const int row = side ? r - P.Mq : r;
    const int n = side ? P.Mq : P.Mr;
    const int pitch = side ? P.pitchT : P.pitchD;
    const float *v = side ? scratch + P.offT + (size_t)row * P.pitchT
                          : scratch + P.offD + (size_t)row * P.pitchD;

    // percentile position, f32 like the oracle (percentile_f32)
    const float kf = (n > 1) ? __fmul_rn((float)(n - 1), kappa) : __fmul_rn((float)n, kappa);
    const float fl = floorf(kf), ce = ceilf(kf);
    int ilo = (int)fl, ihi = (int)ce;
    ilo = ilo < 0 ? 0 : (ilo > n - 1 ? n - 1 : ilo);
    ihi = ihi < 0 ? 0 : (ihi > n - 1 ? n - 1 : ihi);

    // one selection: rank k, plus (interpolating modes) the next order statistic
    int k = ilo;
    if (pct_mode == 3) {
        k = (int)floorf(__fadd_rn(kf, 0.5f));
        k = k > n - 1 ? n - 1 : k;
    }
    const bool interp = (pct_mode == 0 || pct_mode == 1);
    const SelectResult sr = wave_select<V4>(v, n, pitch, k, hist[wave], cand[wave], &counter[wave], lane, interp);
    const float slo = sr.value, nx = sr.next;
    const int cle = sr.cnt_le;
    float eps;
    if (!interp) {
        eps = __builtin_sqrtf(slo);
    } else {
        float shi = slo;
        if (ihi != ilo && cle <= ihi) shi = nx;     // rank ihi is the next distinct value
        const float dlo = __builtin_sqrtf(slo), dhi = __builtin_sqrtf(shi);
        if (pct_mode == 0 && ihi == ilo) {
            eps = dlo;
        } else {
            const float d0 = __fmul_rn(dlo, __fsub_rn(ce, kf));
            const float d1 = __fmul_rn(dhi, __fsub_rn(kf, fl));
            eps = __fadd_rn(d0, d1);
        }
    }
    if (lane == 0) {
        float *X = thr + P.offX;
        const int o = side ? P.pitchT + row : row;
        X[o] = d2_threshold(eps, inclusive);
        X[P.pitchT + P.pitchD + o] = eps;
    }
}

// ------------------------------------------------------------------------------------
// K0b: table of embedded squared norms, built once per (pool, m): for every track, every
// rotation rot = 0..11 and every embedded frame i: xx = tree over m frame norms, each a 12-term
// fmaf chain over the bins in ROTATED order (rotated[c'] = src[(c' - rot) mod 12]) -- the chain
// order is part of the arithmetic spec, so the norm depends on the rotation.  Layout:
// tab[noff[track] + rot * Memb(track) + i].  A pair only picks two rows of it (query unrotated,
// reference rotated by its OTI, or the other way round).
// ------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256) void normtab_kernel(const float *__restrict__ pool,
                                                      const int64_t *__restrict__ toff,
                                                      const int64_t *__restrict__ noff,
                                                      float *__restrict__ tab, int span)
{
    const int track = blockIdx.y, rot = blockIdx.z;
    const int64_t t0 = toff[track];
    const int T = (int)(toff[track + 1] - t0);
    const int Me = T - span;                                   // embedded frames (tau == 1)
    const int i = blockIdx.x * 256 + threadIdx.x;
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
    tab[noff[track] + (int64_t)rot * Me + i] = tree_sum<M>(s);
}

// ------------------------------------------------------------------------------------
// K1': fused band kernel.  One workgroup (8 waves) owns a band of 8 rows of the embedded
// distance matrix; the band is cut into tiles of 64 columns and WAVE w sweeps tiles
// w, w+8, w+16, ... on its own -- no workgroup barrier and no LDS staging of inputs in the sweep:
//   * MFMA operands are plain global loads from the rotated frame pool (rotpool_kernel): one
//     12-byte load per lane per 16-frame tile, already in the rotated chain order of the
//     arithmetic spec; the next tile's operands are in flight while the current one is worked on;
//   * the (8+m-1) x (64+7+m-1) frame Gram is built on the matrix cores
//     (v_mfma_f32_16x16x4_f32, K = 12 in 3 k-steps; row-frame operands stay in registers for
//     the whole band) and parked in the wave's private LDS slab (one 16-byte store per block);
//   * lane c walks the 8 cells (a, c + a), a = 0..7, down one diagonal: m+7 LDS reads give
//     all 8 window sums (doubling-tree subterms are shared between the cells); tile t
//     therefore covers, for band row a, the 64 columns 64 t - 7 + a ...;
//   * distances stay in registers (debug: also to the row-major D2 matrix in HBM).
// After the sweep the 8 waves exchange their pieces through LDS so that wave w holds band
// row w completely (32 values per lane) and runs the exact percentile selection on it
// (wave_select_fast, one histogram pass; wave_select_regs as the generic fallback).
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
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one output float each
    if (idx >= nframes * FROT) return;
    const int64_t f = idx / FROT;
    const int e = (int)(idx - f * FROT);
    const int r = e / NBIN, rem = e - r * NBIN, cls = rem / 3, kb = rem - 3 * cls;
    frot[idx] = pool[f * NBIN + cls + 4 * ((r + kb) % 3)];
}

// Development aid (-DACX_TIMING): per-phase shader-clock totals of band_kernel, summed over all
// waves into acx_tim[] (slot 31 = number of waves).  Not compiled into the product library.
#ifdef ACX_TIMING
static __device__ unsigned long long acx_tim[32];
#define ACX_T(k) do { tstamp[k] = __builtin_readcyclecounter(); } while (0)
#else
#define ACX_T(k) do { } while (0)
#endif

// Host-side launcher of band_kernel: its own translation unit (acx_band.hip), so that the kernel
// can be compiled with the scheduling strategy that suits it without touching the others.
struct BandLaunch {
    hipStream_t stream;
    const float *frot;
    const int64_t *toff;
    const float *normtab;
    const int64_t *noff;
    float *scratch, *thr;
    unsigned long long *bits;
    float kappa;
    int pct_mode, inclusive, oti_target;
};
// role 1 / 0 over B pairs of one size class; false when the stack size m has no instantiation
bool launch_band_kernel(const BandLaunch &L, int m, const struct PairDesc *dpd, int B, int maxRows, int maxCols, int role,
                        int write_d2);
#ifdef ACX_TIMING
hipError_t band_timing(unsigned long long *out32, int reset);      // development builds: per-phase clock totals
#endif

template <int M>
struct BandGeom {
    static constexpr int NRT = (BAND + M - 1 + 15) / 16;            // 16-row MFMA tiles of row frames
    static constexpr int NCT = (64 + BAND - 1 + M - 1 + 15) / 16;   // 16-col MFMA tiles of column frames
    static constexpr int AROWS = 16 * NRT;
    static constexpr int BW = 16 * NCT;
    static constexpr int SP = BW + 4;                              // S pitch: 16-byte aligned rows; 84 % 32 = 20 keeps the 16-byte tile stores conflict-free
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

// (short-row variants: 6 waves / SIMD = 3 workgroups per CU; ACX_OCC6_ALL is a development switch)
#ifdef ACX_OCC6_ALL
#define ACX_OCC6(M_, V4_) ((V4_) <= 4)
#else
#define ACX_OCC6(M_, V4_) ((V4_) <= 4 && (M_) <= 9)      /* (m >= 10 needs two MFMA row tiles: LDS-limited anyway) */
#endif
template <int M, int V4, int ROLE>
__global__ __launch_bounds__(BAND_THREADS, (ACX_OCC6(M, V4) ? 6 : 4)) void band_kernel(const float *__restrict__ frot,
                                                            const int64_t *__restrict__ toff,
                                                            const float *__restrict__ normtab,
                                                            const int64_t *__restrict__ noff,
                                                            const PairDesc *__restrict__ pd,
                                                            float *__restrict__ scratch,
                                                            float *__restrict__ thr,
                                                            unsigned long long *__restrict__ bits,
                                                            float kappa, int pct_mode, int inclusive, int oti_target,
                                                            int write_d2)
{
    using G = BandGeom<M>;
    constexpr int role = ROLE;           // 1: rows = reference frames (column thresholds); 0: rows = query frames
    constexpr int NV = 4 * V4;           // values per lane of a complete row
    constexpr int NSTEP = NV / 8;        // tiles per wave
    constexpr int LNP = NV + 4;          // floats per owner lane in an exchange row: its NV positions + 16 bytes of pad
    constexpr int ROWP = 64 * LNP;       // exchange pitch (floats)
    // Fast selection: FBINS counters = FBINS / FCOPIES bins x FCOPIES copies.  (Copies spread
    // same-address atomics -- they paid while one instruction handled 64 NEIGHBOURING columns; with
    // the position-order rows below one copy and twice the bins measure best on every workload.)
    // For rows of >= 1024 positions the histogram lives in the upper part of the wave's OWN
    // exchange row, free once the row sits in registers; the shortest rows (2 KB) keep a separate
    // area behind the exchange rows.
#ifdef ACX_FBINS
    constexpr int FBINS = ACX_FBINS;
#else
    constexpr int FBINS = NV >= 32 ? 1024 : 512;
#endif
#ifdef ACX_FCOPIES
    constexpr int FCOPIES = ACX_FCOPIES;
#else
    constexpr int FCOPIES = 1;
#endif
    constexpr bool HIST_IN_ROW = ROWP >= 64 + 2 * FBINS;        // room for an FBINS-aligned block behind the 64 candidate slots
    constexpr int GBINS = 32 * NV;                              // bins of the generic (narrowing) selection
    constexpr int SWEEP_FLOATS = 8 * G::AROWS * G::SP;          // one Gram tile per wave
    constexpr int TAIL_FLOATS = BAND * ROWP + (HIST_IN_ROW ? 0 : 8 * FBINS);
#ifdef ACX_LDS_PAD      /* experiment: force one workgroup per CU */
    constexpr int LDS_FLOATS = 24 * 1024;
#else
    constexpr int LDS_FLOATS = SWEEP_FLOATS > TAIL_FLOATS ? SWEEP_FLOATS : TAIL_FLOATS;
#endif
    static_assert(SelGeom<GBINS>::SLOTS + 64 + 4 <= ROWP, "generic selection must fit the wave's own exchange row");
    static_assert(HIST_IN_ROW || (BAND * ROWP) % FBINS == 0, "fast histograms must be aligned to their size");
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
    const float *nrow = normtab + noff[role ? P.r : P.q] + (int64_t)rota * MA;
    const float *ncol = normtab + noff[role ? P.q : P.r] + (int64_t)rotb * MB;
    const float INF = __builtin_inff();
    // Cells outside the matrix travel through the exchange as -1: distances are >= +0, so a negative
    // pad is the largest UNSIGNED and the smallest SIGNED bit pattern -- the selection's integer min /
    // max skip it for free (a +inf pad needed a bias add per value for the max)
    const float PADV = -1.0f;
#ifdef ACX_TIMING
    unsigned long long tstamp[16];
    unsigned long long tsub[4] = {0, 0, 0, 0};
    for (int q = 0; q < 16; ++q) tstamp[q] = 0;
#endif
    ACX_T(0);

    // ---- MFMA operands come straight from the rotated frame pool (frot, see rotpool_kernel):
    // frame f holds, for each rotation r = 0..2 and residue class cls = 0..3, the three bins
    // cls + 4 ((r + kb) mod 3), kb = 0..2, contiguously.  The lane that feeds k-position lk of
    // the MFMA chain needs rotated bin 4 kb + lk = source bin (4 kb + lk - rot) mod 12, i.e. with
    // c0 = (lk - rot) mod 12 exactly the triple (r = c0 / 4, cls = c0 % 4): ONE 12-byte load per
    // 16-frame tile, already in chain order.  No LDS staging, no LDS-DMA, no barrier.
    const int lr = lane & 15, lk = lane >> 4;
    int c0a = lk - rota; if (c0a < 0) c0a += NBIN;
    int c0b = lk - rotb; if (c0b < 0) c0b += NBIN;
    const int offA = (c0a >> 2) * NBIN + (c0a & 3) * 3;
    const int offB = (c0b >> 2) * NBIN + (c0b & 3) * 3;
    const float *fra = frot + toff[role ? P.r : P.q] * FROT + offA;
    const float *frb = frot + toff[role ? P.q : P.r] * FROT + offB;
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    typedef f32x3 f32x3_u __attribute__((aligned(4)));
    float areg[G::NRT][3];
#pragma unroll
    for (int ta = 0; ta < G::NRT; ++ta) {
        int f = i0 + 16 * ta + lr;
        f = f > TA - 1 ? TA - 1 : f;           // rows beyond the matrix are masked below
        const f32x3 v = *reinterpret_cast<const f32x3_u *>(fra + (size_t)f * FROT);
        areg[ta][0] = v.x; areg[ta][1] = v.y; areg[ta][2] = v.z;
    }
    float xrow[BAND];
#pragma unroll
    for (int a = 0; a < BAND; ++a) xrow[a] = (i0 + a < MA) ? nrow[i0 + a] : 0.0f;
    float *Sw = smem + wave * (G::AROWS * G::SP);               // this wave's Gram tile, [row frame][column frame]

    const int ntiles = (MB + BAND - 1 + 63) / 64;      // <= NV by dispatch
    typedef float BvT[G::NCT][3];
    typedef f32x4 AccT[G::NRT][G::NCT];
    // column-frame operands of a tile (frames 64 tile - 7 ... + BW)
    // (tb0: first 16-frame block wanted -- a wave's second and later tiles inherit their first HB
    // blocks from the tile before, see the sweep)
    auto load_operands = [&](int tile, BvT &bv, auto tb0_tag) {
        constexpr int tb0 = decltype(tb0_tag)::value;
        const int base = 64 * tile - (BAND - 1);
        if (base >= 0 && base + G::BW <= TB) {                  // wave-uniform: all frames exist
            const float *p = frb + (ptrdiff_t)(base + lr) * FROT;
#pragma unroll
            for (int tb = tb0; tb < G::NCT; ++tb) {
                const f32x3 v = *reinterpret_cast<const f32x3_u *>(p + 16 * FROT * tb);
                bv[tb][0] = v.x; bv[tb][1] = v.y; bv[tb][2] = v.z;
            }
        } else {                                                // clamp: those cells are masked anyway
#pragma unroll
            for (int tb = tb0; tb < G::NCT; ++tb) {
                int f = base + 16 * tb + lr;
                f = f < 0 ? 0 : (f > TB - 1 ? TB - 1 : f);
                const f32x3 v = *reinterpret_cast<const f32x3_u *>(frb + (ptrdiff_t)f * FROT);
                bv[tb][0] = v.x; bv[tb][1] = v.y; bv[tb][2] = v.z;
            }
        }
    };
    // embedded column norms of the lane's 8 cells
    auto load_norms = [&](int tile, float (&yv)[BAND]) {
        const int base = 64 * tile - (BAND - 1);
        if (base >= 0 && base + 64 + BAND - 1 <= MB) {
            // the lane's 8 consecutive norms as two (4-byte aligned) 16-byte loads
            typedef float f32x4n __attribute__((ext_vector_type(4), aligned(4)));
            const f32x4n *p = reinterpret_cast<const f32x4n *>(ncol + base + lane);
            const f32x4n v0 = p[0], v1 = p[1];
            yv[0] = v0.x; yv[1] = v0.y; yv[2] = v0.z; yv[3] = v0.w;
            yv[4] = v1.x; yv[5] = v1.y; yv[6] = v1.z; yv[7] = v1.w;
            static_assert(BAND == 8, "two 16-byte loads cover the band's 8 norms");
        } else {
#pragma unroll
            for (int a = 0; a < BAND; ++a) {
                int j = base + lane + a;
                j = j < 0 ? 0 : (j > MB - 1 ? MB - 1 : j);
                yv[a] = ncol[j];
            }
        }
    };
    // frame Gram on the matrix cores.  The COLUMN frames are the MFMA's row operand, so a lane ends
    // up with four consecutive column frames of one row frame: one 16-byte LDS store per 16x16 tile
    // (products commute, the k order is unchanged: same bits).  Chains interleaved k-step-major so
    // that no MFMA waits on its predecessor.
    auto gram = [&](const BvT &bv, AccT &acc, auto tb0_tag) {
        constexpr int tb0 = decltype(tb0_tag)::value;
#pragma unroll
        for (int ta = 0; ta < G::NRT; ++ta)
#pragma unroll
            for (int tb = tb0; tb < G::NCT; ++tb) acc[ta][tb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int ta = 0; ta < G::NRT; ++ta)
#pragma unroll
                for (int tb = tb0; tb < G::NCT; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[tb][kb], areg[ta][kb], acc[ta][tb], 0, 0, 0);
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
    ACX_T(1);

    const int pitchD = P.pitchD;
    float *D = scratch + P.offD + (size_t)i0 * pitchD;
    float xv[BAND][NSTEP];      // (every element is written below: cells of a tile, or +inf for a tile past the row)

    auto keep = [&](auto st_tag, int tile, const float (&dv)[BAND]) {     // cells -> xv (+ debug D2)
        constexpr int st = decltype(st_tag)::value;
        const int base = 64 * tile - (BAND - 1);
        const int j0 = base + lane;                       // column of the lane's first cell
        const bool interior = base >= 0 && base + 64 + BAND - 1 <= MB && i0 + BAND <= MA;   // wave-uniform
        if (interior) {
#pragma unroll
            for (int a = 0; a < BAND; ++a) xv[a][st] = dv[a];
#ifndef ACX_ABL_NOSTORE
            if (write_d2) {
                float *Dl = D + j0;
#pragma unroll
                for (int a = 0; a < BAND; ++a) Dl[a * pitchD + a] = dv[a];
            }
#endif
        } else {
#pragma unroll
            for (int a = 0; a < BAND; ++a) {
                const int j = j0 + a;
                const bool rowok = (i0 + a) < MA;
                const bool ok = rowok && j >= 0 && j < MB;
                const float v = ok ? dv[a] : INF;
                xv[a][st] = ok ? dv[a] : PADV;
                if (write_d2 && rowok && j >= 0 && j < pitchD) D[a * pitchD + j] = v;
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
    BvT bvbuf[2];
    f32x4 halo[G::NRT][HB];
    if (wave * cpw < ntiles) load_operands(wave * cpw, bvbuf[0], std::integral_constant<int, 0>());
    static_for<0, NSTEP>([&](auto st_tag) {
        constexpr int st = decltype(st_tag)::value;
        constexpr int tb0 = st == 0 ? 0 : HB;            // first block this step computes
        const int tile = tile_of(st);
        if (st < cpw && tile < ntiles) {      // wave-uniform
            float yv[BAND];
            load_norms(tile, yv);
            if (st + 1 < NSTEP && st + 1 < cpw && tile + 1 < ntiles)
                load_operands(tile + 1, bvbuf[(st + 1) & 1], std::integral_constant<int, HB>());
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
                gram(bvbuf[st & 1], acc, std::integral_constant<int, tb0>());
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
    ACX_T(2);
    // debug / v1 consumers: +inf into the pad columns [MB, pitchD) of the band's rows
    if (write_d2) {
        const int npad = pitchD - MB;
        for (int idx = tid; idx < BAND * npad; idx += BAND_THREADS) {
            const int a = idx / npad, j = MB + idx - a * npad;
            if (i0 + a < MA) D[a * pitchD + j] = INF;
        }
    }
    __syncthreads();     // all slabs dead -> reuse LDS as the exchange rows + the fast histograms
    ACX_T(3);
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
    // this wave's fast histogram: inside its own exchange row (the first FBINS-aligned block behind
    // the 64 candidate slots), or in the separate area behind the rows
    const int hist_off = HIST_IN_ROW ? ((wave * ROWP + 64 + FBINS - 1) & ~(FBINS - 1)) : BAND * ROWP + wave * FBINS;
    if constexpr (!HIST_IN_ROW) {   // zero it before the barrier: nobody else touches that area
        float *h = smem + hist_off;
#pragma unroll
        for (int q = 0; q < FBINS / 256; ++q) *reinterpret_cast<float4 *>(h + 256 * q + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    ACX_T(4);
    __syncthreads();
    ACX_T(5);
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

    // ---- exact percentile selection: wave w owns band row w
    const int row = i0 + wave;
    if (row >= MA) return;
#ifdef ACX_ABL_NOSELECT
    if (lane == 0) { float acc_ = 0; for (int e = 0; e < NV; ++e) acc_ += xr[e]; thr[P.offX + (role ? P.pitchT + row : row)] = acc_; }
    return;
#endif
    const int n = MB;
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
    float *myrow = smem + wave * ROWP;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the row has left LDS
#if defined(ACX_EXTRA_VALU) || defined(ACX_EXTRA_LDS) || defined(ACX_EXTRA_MFMA) || defined(ACX_EXTRA_VALU4)
    {   // sensitivity experiment (development): extra independent work of one kind per wave
        float e0 = xr[0], e1 = xr[1], e2 = xr[2], e3 = xr[3];
#ifdef ACX_EXTRA_VALU
#pragma unroll
        for (int q = 0; q < ACX_EXTRA_VALU / 4; ++q)
            asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(xr[4]));
#endif
#ifdef ACX_EXTRA_VALU4
#pragma unroll
        for (int q = 0; q < ACX_EXTRA_VALU4 / 4; ++q)
            asm volatile("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(xr[4]));
#endif
#ifdef ACX_EXTRA_LDS
        {
            const unsigned la = (unsigned)(uintptr_t)(lds_void *)(myrow) + 4u * lane;
            unsigned t0, t1, t2, t3;
#pragma unroll
            for (int q = 0; q < ACX_EXTRA_LDS / 4; ++q) {
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                             : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(la) : "memory");
                e0 += __uint_as_float(t0 ^ t1 ^ t2 ^ t3) * 0.0f;
            }
        }
#endif
#ifdef ACX_EXTRA_MFMA
        {
            f32x4 ac = {0.f, 0.f, 0.f, 0.f}, ad = ac;
#pragma unroll
            for (int q = 0; q < ACX_EXTRA_MFMA / 2; ++q) {
                ac = __builtin_amdgcn_mfma_f32_16x16x4f32(e0, e1, ac, 0, 0, 0);
                ad = __builtin_amdgcn_mfma_f32_16x16x4f32(e2, e3, ad, 0, 0, 0);
            }
            e0 += (ac[0] + ad[1]) * 0.0f;
        }
#endif
        xr[0] += (e0 + e1 + e2 + e3) * 0.0f - (xr[0] + xr[1] + xr[2] + xr[3]) * 0.0f;
    }
#endif
    ACX_T(6);
    float slo, shi;
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned hist_addr = (unsigned)(uintptr_t)(lds_void *)(smem + hist_off);
    if constexpr (HIST_IN_ROW) {    // the row has left LDS: part of it becomes the zeroed histogram
        float *h = smem + hist_off;
#pragma unroll
        for (int q = 0; q < FBINS / 256; ++q) *reinterpret_cast<float4 *>(h + 256 * q + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
        wave_lds_fence();
    }
    bool done = false;
    const int end_valid = MB + (BAND - 1) - wave;                       // positions [7 - wave, end_valid) are cells
    const bool lane_has_data = lane * NV < end_valid;                   // first position of the lane is a cell or a low pad
    // pads inside the lanes that take part in the histogram: the low ones of lane 0 and the tail of the
    // last lane with cells; they sit in bin 0 and rank below every cell
    const int npadc = ((BAND - 1) - wave) + (((end_valid + NV - 1) / NV) * NV - end_valid);
#ifndef ACX_NO_FASTSEL
#ifdef ACX_TIMING
    unsigned long long tsel[6] = {0, 0, 0, 0, 0, 0};
    done = wave_select_fast<NV, FBINS, FCOPIES, true>(xr, k + npadc, interp && ihi != ilo, hist_addr, myrow, lane, slo, shi, tsel, lane_has_data);
    if (lane == 0 && done && tsel[5] && (blockIdx.x & 31) == 5) for (int q = 0; q < 5; ++q) atomicAdd(&acx_tim[20 + q], tsel[q + 1] - tsel[q]);
    if (lane == 0 && (blockIdx.x & 31) == 5) atomicAdd(&acx_tim[done ? 26 : 25], 1ull);      // fast-path hits / fallbacks
#else
    done = wave_select_fast<NV, FBINS, FCOPIES, true>(xr, k + npadc, interp && ihi != ilo, hist_addr, myrow, lane, slo, shi, nullptr, lane_has_data);
#endif
#endif
    if (!done) {
        unsigned *ghist = reinterpret_cast<unsigned *>(myrow) + 64;
        unsigned *counter = reinterpret_cast<unsigned *>(myrow) + 64 + SelGeom<GBINS>::SLOTS;
        float xi[NV];                                   // the generic selection wants its pads at +inf
#pragma unroll
        for (int t = 0; t < NV; ++t) xi[t] = xr[t] < 0.0f ? INF : xr[t];
        const SelectResult sr = wave_select_regs<NV, GBINS>(xi, k, ghist, myrow, counter, lane, interp);
        slo = sr.value;
        shi = (interp && ihi != ilo && sr.cnt_le <= ihi) ? sr.next : sr.value;     // rank ihi is the next distinct value
    }
#if defined(ACX_ABL_STAGE) && ACX_ABL_STAGE == 1
    if (lane == 0) thr[P.offX + (role ? P.pitchT + row : row)] = slo + shi;
    return;
#endif
    ACX_T(7);
    const float eps = percentile_eps2(slo, shi, pct_mode, ilo, ihi, kf, fl, ce);
    const float thr_row = d2_threshold(eps, inclusive);
#ifdef ACX_TIMING
    asm volatile("" :: "v"(thr_row));
#endif
    ACX_T(8);
    float *X = thr + P.offX;
    if (lane == 0) {
        const int o = role ? P.pitchT + row : row;
        X[o] = thr_row;
        X[P.pitchT + P.pitchD + o] = eps;
    }
#if defined(ACX_ABL_STAGE) && ACX_ABL_STAGE == 2
    return;
#endif
    // ---- role 0 (the column thresholds of the pair are already there): binarise the row the
    // wave still holds in registers and emit it as a bitmap -- bit p of the row = position p =
    // column p - 7 + (row & 7).  256 bytes per row instead of 8 KB of f32.  A lane owns NV
    // consecutive positions, i.e. NV consecutive bits: R = [d2 <= min(thr_row, thr_col)] is shifted
    // into the lane's own word bit by bit (compare -> carry -> add-with-carry), no cross-lane traffic.
    if (role == 0 && bits) {
        // column thresholds (d2 domain): NV consecutive floats per lane, no bounds check (columns
        // -7 .. 64 ntiles + 63 of the threshold arena are always inside the pair's arena)
        const float *tc = X + P.pitchT + (lane * NV + wave - (BAND - 1));
        typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
        float tcv[NV];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const f32x4_u v = *reinterpret_cast<const f32x4_u *>(tc + 4 * j);
            tcv[4 * j + 0] = v.x; tcv[4 * j + 1] = v.y; tcv[4 * j + 2] = v.z; tcv[4 * j + 3] = v.w;
        }
        // positions of this lane whose column exists: t in [lo, hi)
        int lo = (BAND - 1) - wave - lane * NV, hi = MB + (BAND - 1) - wave - lane * NV;
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
        if constexpr (NV == 16) {
            acc |= (unsigned)__shfl_down((int)acc, 1, 64) << 16;
        } else if constexpr (NV == 8) {
            acc |= (unsigned)__shfl_down((int)acc, 1, 64) << 8;
            acc |= (unsigned)__shfl_down((int)acc, 2, 64) << 16;
        }
        constexpr int LPD = 32 / NV;                                    // lanes per dword
        unsigned *rowbits = reinterpret_cast<unsigned *>(bits + P.offT + (size_t)row * P.nw);
        const int ndw = 2 * P.nw, d = lane / LPD;
        if ((lane & (LPD - 1)) == 0 && d < ndw) rowbits[d] = acc;
        for (int z = 2 * NV + lane; z < ndw; z += 64) rowbits[z] = 0u;  // words beyond this size class
    }
    ACX_T(9);
#ifdef ACX_TIMING
    if (lane == 0 && (blockIdx.x & 31) == 5) {
        for (int q = 0; q < 9; ++q) atomicAdd(&acx_tim[q], tstamp[q + 1] - tstamp[q]);
        for (int q = 0; q < 3; ++q) atomicAdd(&acx_tim[16 + q], tsub[q]);
        atomicAdd(&acx_tim[31], 1ull);
    }
#endif
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
        const unsigned w = wraw & colmask;
        float l1a = wave_shfl(QA[CPL - 1], prev), l1b = wave_shfl(QA[CPL - 2], prev), l2a = wave_shfl(QB[CPL - 1], prev);
        float p1a = 0.f, p1b = 0.f, p2a = 0.f;
        if constexpr (!EQG) {
            p1a = wave_shfl(PA[CPL - 1], prev); p1b = wave_shfl(PA[CPL - 2], prev); p2a = wave_shfl(PB[CPL - 1], prev);
        }
        if (lane == 0) { l1a = 0.f; l1b = 0.f; l2a = 0.f; p1a = 0.f; p1b = 0.f; p2a = 0.f; }
        unsigned wleft = 0u;                                    // bit e = R[i][j-1]
        if constexpr (DMAX) {
            unsigned carry = (unsigned)__shfl((int)((wraw >> (CPL - 1)) & 1u), prev, 64);
            if (lane == 0) carry = 0u;
            wleft = (wraw << 1) | carry;
        }
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
            float vgap;
            if constexpr (EQG) {
                vgap = fmaxf(mx - go, 0.0f);
            } else {
                const float a2 = (e >= 1) ? PA[e - 1] : p1a;
                float a3 = (e >= 1) ? PB[e - 1] : p2a;
                float a4 = (e >= 2) ? PA[e - 2] : (e == 1 ? p1a : p1b);
                if constexpr (DMAX) { a3 += x3; a4 += x4; }
                vgap = fmaxf(fmaxf(fmaxf(a2, a3), a4), 0.0f);
            }
            float q = r ? (mx + 1.0f) : vgap;
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
// K3: Qmax / Dmax row sweep, one wave per pair.  Lane owns NG groups of 8 contiguous
// columns: group g = columns [512 g + 8 lane, +8).  Q rows i-1 / i-2 live in registers;
// the row in flight is updated in place (descending column order).
// EQG: gamma_o == gamma_e (the default) -> max(a-g, b-g, c-g) == max(a,b,c)-g exactly.
// ------------------------------------------------------------------------------------
template <int NG>
__device__ __forceinline__ void qmax_load_row(float (&buf)[NG][8], const float *__restrict__ D,
                                              int pitch, int row, int nrows, int lane)
{
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int col = 512 * g + 8 * lane;
        float4 t0 = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff());
        float4 t1 = t0;
        if (row < nrows && col < pitch) {
            const float4 *p = reinterpret_cast<const float4 *>(D + (size_t)row * pitch + col);
            t0 = p[0];
            t1 = p[1];
        }
        buf[g][0] = t0.x; buf[g][1] = t0.y; buf[g][2] = t0.z; buf[g][3] = t0.w;
        buf[g][4] = t1.x; buf[g][5] = t1.y; buf[g][6] = t1.z; buf[g][7] = t1.w;
    }
}

// One DP row.  P1/P2: Q rows i-1 / i-2 (P2 is overwritten with row i).
template <int NG, bool EQG>
__device__ __forceinline__ void qmax_row(const float (&buf)[NG][8],
                                         float (&Q1)[NG][8], float (&Q2)[NG][8],
                                         float (&Pn1)[EQG ? 1 : NG][8], float (&Pn2)[EQG ? 1 : NG][8],
                                         const float (&xc)[NG][8], float xrow,
                                         float go, float ge, int lane, float &best)
{
    const int prev = (lane + 63) & 63;
    // values of the left neighbour columns (previous lane, or lane 63 of the previous group)
    float l1a[NG], l1b[NG], l2a[NG];     // Q1[j0-1], Q1[j0-2], Q2[j0-1]
    float p1a[NG], p1b[NG], p2a[NG];     // penalised versions (gammas differ)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        float a = wave_shfl(Q1[g][7], prev), b = wave_shfl(Q1[g][6], prev), c = wave_shfl(Q2[g][7], prev);
        l1a[g] = a; l1b[g] = b; l2a[g] = c;
        if constexpr (!EQG) {
            p1a[g] = wave_shfl(Pn1[g][7], prev);
            p1b[g] = wave_shfl(Pn1[g][6], prev);
            p2a[g] = wave_shfl(Pn2[g][7], prev);
        }
    }
    // lane 0 takes the wrapped values from the previous group (or zeros at the matrix edge)
#pragma unroll
    for (int g = NG - 1; g >= 0; --g) {
        if (lane == 0) {
            l1a[g] = g > 0 ? l1a[g - 1] : 0.0f;
            l1b[g] = g > 0 ? l1b[g - 1] : 0.0f;
            l2a[g] = g > 0 ? l2a[g - 1] : 0.0f;
            if constexpr (!EQG) {
                p1a[g] = g > 0 ? p1a[g - 1] : 0.0f;
                p1b[g] = g > 0 ? p1b[g - 1] : 0.0f;
                p2a[g] = g > 0 ? p2a[g - 1] : 0.0f;
            }
        }
    }
    // NOTE on the lane-0 fix-up above: for lane 0 the shuffled value is lane 63's register of
    // the SAME group; the wanted one is lane 63's register of group g-1, which is what
    // l1a[g-1] holds for lane 0 BEFORE its own fix-up -- hence the descending g order.

#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float (&d)[8] = buf[g];
        float qn[8], pn[8];
        bool rr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) rr[e] = d[e] <= fminf(xrow, xc[g][e]);
#pragma unroll
        for (int e = 7; e >= 0; --e) {
            const float c2 = (e >= 1) ? Q1[g][e - 1] : l1a[g];                       // (i-1, j-1)
            const float c3 = (e >= 1) ? Q2[g][e - 1] : l2a[g];                             // (i-2, j-1)
            const float c4 = (e >= 2) ? Q1[g][e - 2] : (e == 1 ? l1a[g] : l1b[g]);         // (i-1, j-2)
            float mx = fmaxf(fmaxf(c2, c3), c4);
            float vmatch = mx + 1.0f;
            float vgap;
            if constexpr (EQG) {
                vgap = fmaxf(mx - go, 0.0f);
            } else {
                const float a2 = (e >= 1) ? Pn1[g][e - 1] : p1a[g];
                float a3 = (e >= 1) ? Pn2[g][e - 1] : p2a[g];
                float a4 = (e >= 2) ? Pn1[g][e - 2] : (e == 1 ? p1a[g] : p1b[g]);
                vgap = fmaxf(fmaxf(fmaxf(a2, a3), a4), 0.0f);
            }
            qn[e] = rr[e] ? vmatch : vgap;
            pn[e] = qn[e] - (rr[e] ? go : ge);
        }
        if (g == 0 && lane == 0) { qn[0] = 0.0f; qn[1] = 0.0f; pn[0] = 0.0f; pn[1] = 0.0f; }   // first two columns stay 0
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            Q2[g][e] = qn[e];
            if constexpr (!EQG) Pn2[g][e] = pn[e];
            best = fmaxf(best, qn[e]);
        }
    }
}

template <int NG, bool EQG>
__global__ __launch_bounds__(64) void qmax_kernel(const PairDesc *__restrict__ pd,
                                                  const float *__restrict__ scratch,
                                                  const float *__restrict__ thr,
                                                  float *__restrict__ out,
                                                  float go, float ge, int dp_start)
{
    const int lane = threadIdx.x;
    const PairDesc P = pd[blockIdx.x];
    int Me = P.Mq, Ne = P.Mr;
    if (dp_start == 3) { Me -= 1; Ne -= 1; }
    const float *D = scratch + P.offD;
    const int pitch = P.pitchD;
    const float *xr = thr + P.offX;
    const float *xcp = xr + P.pitchT;

    float xc[NG][8];
    float QA[NG][8], QB[NG][8];
    float PA[EQG ? 1 : NG][8], PB[EQG ? 1 : NG][8];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 512 * g + 8 * lane + e;
            xc[g][e] = (j < Ne) ? xcp[j] : -1.0f;    // -1: never recurrent (d2 >= 0)
            QA[g][e] = 0.0f;
            QB[g][e] = 0.0f;
            if constexpr (!EQG) { PA[g][e] = 0.0f; PB[g][e] = 0.0f; }
        }
    float best = 0.0f;

    float b0[NG][8], b1[NG][8], b2[NG][8];
    qmax_load_row<NG>(b0, D, pitch, 2, Me, lane);
    qmax_load_row<NG>(b1, D, pitch, 3, Me, lane);
    float xrv = 0.0f;
    for (int i = 2; i < Me; i += 6) {
        // thresholds of rows i .. i+5 (lane t holds row i + t)
        xrv = (i + lane < Me && lane < 6) ? xr[i + lane] : -1.0f;
#define ACX_QSTEP(S, BUF, NEXT, Q1, Q2, P1, P2)                                                     \
        if (i + S < Me) {                                                                           \
            qmax_load_row<NG>(NEXT, D, pitch, i + S + 2, Me, lane);                                 \
            const float xrow = __shfl(xrv, S, 64);                                                  \
            qmax_row<NG, EQG>(BUF, Q1, Q2, P1, P2, xc, xrow, go, ge, lane, best);              \
        }
        // row i+S reads Q1 = row i+S-1, Q2 = row i+S-2 and overwrites Q2
        ACX_QSTEP(0, b0, b2, QA, QB, PA, PB)   // QA = row i-1, QB = row i-2 -> QB = row i
        ACX_QSTEP(1, b1, b0, QB, QA, PB, PA)
        ACX_QSTEP(2, b2, b1, QA, QB, PA, PB)
        ACX_QSTEP(3, b0, b2, QB, QA, PB, PA)
        ACX_QSTEP(4, b1, b0, QA, QB, PA, PB)
        ACX_QSTEP(5, b2, b1, QB, QA, PB, PA)
#undef ACX_QSTEP
    }
    best = wave_max(best);
    if (lane == 0) out[blockIdx.x] = best;
}

static __global__ void sqrt_probe_kernel(const float *in, float *out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_sqrtf(in[i]);
}

}  // namespace acx
