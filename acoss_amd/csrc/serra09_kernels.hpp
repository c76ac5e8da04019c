// Serra09 device kernels for gfx950 (MI355X / CDNA4).  Wave = 64 lanes.
//
// Per-pair chain (reference call site acoss/algorithms/rqa_serra09.py:55-69, arithmetic
// spec in DESIGN.md / oracle/acx_oracle.c):
//
//   K0 oti_kernel      12-bin optimal transposition index per pair
//   K1 csm_tile_kernel 64x64 tiles of the embedded squared-distance matrix:
//                      frame Gram on the matrix cores (v_mfma_f32_16x16x4_f32, K = 12 =
//                      3 k-steps, exact f32 == fmaf chain), S tile staged in LDS, m-term
//                      diagonal doubling-tree window sums, writes D2 (row-major) and its
//                      transpose D2T (so column thresholds are row selections)
//   K2 rowsel_kernel   one wave per row: exact order statistics by histogram radix
//                      narrowing in LDS -> kappa-percentile threshold, moved to the d2
//                      domain
//   K3 qmax_kernel     one wave per pair, row sweep: binarise on the fly
//                      (d2 <= min(thr_row, thr_col)) and run the Qmax recurrence with the
//                      two previous rows in registers; neighbours j-1/j-2 across lanes by
//                      wave rotate
//
// Everything is f32; the operation ORDER is part of the spec (bit-exact parity with the
// oracle), so this file is compiled with -ffp-contract=off and uses explicit fmaf only
// where the spec says fma.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acx {

constexpr int NBIN = 12;       // chroma bins
constexpr int TILE = 64;       // output tile edge of K1
constexpr int STILE = 80;      // S tile edge (TILE + 16 halo): supports m <= 17
constexpr int SPITCH = 81;
constexpr int SEL_BINS = 2048; // histogram bins of K2
constexpr int MAX_M = 16;

struct PairDesc {
    int32_t q, r;          // track indices (query, reference)
    int32_t Tq, Tr;        // pooled lengths
    int32_t Mq, Mr;        // embedded lengths (rows, cols of the matrix)
    int32_t oti;           // filled by K0
    int32_t pitchD;        // row pitch of D2  (floats, multiple of 64, >= Mr)
    int32_t pitchT;        // row pitch of D2T (floats, multiple of 64, >= Mq)
    int32_t pad_;
    int64_t offD, offT;    // float offsets into the scratch arena
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
__global__ void oti_kernel(PairDesc *pd, int B, const float *__restrict__ gch, int oti_on, int oti_target)
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

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// histogram slot of logical bin b: one pad word per 32 bins, so that neighbouring bins sit
// in neighbouring banks (atomics) AND the scan, where lane L reads bins 32L..32L+31, is
// conflict-free ((33 L + e) mod 32 distinct over L).
constexpr int SEL_SLOTS = SEL_BINS + SEL_BINS / 32;
__device__ __forceinline__ int hslot(int b) { return b + (b >> 5); }

struct SelectResult { float value; int cnt_le; float next; };

// k-th smallest (0-based) of a row of n <= 64 * 4 * V4 floats -- exact.  The row is read
// ONCE with V4 back-to-back 16-byte loads per lane (element 256 q + 4 lane + e) and then
// lives in registers; every later pass (range, histogram, gather, next-greater) is
// register + LDS only.  Iterative narrowing: histogram the active value range into
// SEL_BINS linear bins, descend into the bin holding rank k, until it holds <= 64
// elements, which are ranked directly.  If want_next, also returns #(v <= result) and
// min{v > result} (+inf if none).
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
        const float scale = (float)SEL_BINS / (mx - mn);
        for (int b = lane * 4; b < SEL_SLOTS; b += 256)
            *reinterpret_cast<uint4 *>(hist + b) = make_uint4(0, 0, 0, 0);
        if (lane == 0) *counter = 0u;
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if (x[t] >= mn && x[t] <= mx) {
                int b = (int)((x[t] - mn) * scale);
                b = b > SEL_BINS - 1 ? SEL_BINS - 1 : b;
                __hip_atomic_fetch_add(&hist[hslot(b)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        wave_lds_fence();
        // scan: lane owns bins [32*lane, 32*lane+32)
        int hv[32];
        int lsum = 0;
#pragma unroll
        for (int e = 0; e < 32; ++e) { hv[e] = (int)hist[33 * lane + e]; lsum += hv[e]; }
        int incl = lsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        const int target = k - below;                 // rank inside the active set
        const unsigned long long m = __ballot(incl > target);
        const int L = __ffsll((long long)m) - 1;      // first lane whose inclusive sum exceeds target
        const int excl = incl - lsum;
        int binsel = 0, cum = 0, cnt = 0;
        if (lane == L) {
            int run = excl;
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                if (cnt == 0 && run + hv[e] > target) { binsel = 32 * lane + e; cum = run; cnt = hv[e]; }
                run += hv[e];
            }
        }
        binsel = __shfl(binsel, L, 64);
        cum = __shfl(cum, L, 64);
        cnt = __shfl(cnt, L, 64);
        if (cnt <= 64) {
            // append the bin's elements (<= 64) to cand[] (order irrelevant), rank, pick
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                if (x[t] >= mn && x[t] <= mx) {
                    int b = (int)((x[t] - mn) * scale);
                    b = b > SEL_BINS - 1 ? SEL_BINS - 1 : b;
                    if (b == binsel) {
                        unsigned pos = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        cand[pos & 63u] = x[t];
                    }
                }
            }
            wave_lds_fence();
            const float mine = (lane < cnt) ? cand[lane] : INF;
            int rank = 0;
            for (int t = 0; t < cnt; ++t) {
                const float o = cand[t];
                rank += (o < mine || (o == mine && t < lane)) ? 1 : 0;
            }
            const int want = target - cum;
            const unsigned long long hit = __ballot(lane < cnt && rank == want);
            const int src = __ffsll((long long)hit) - 1;
            result = __shfl(mine, src, 64);
            wave_lds_fence();
            break;
        }
        // narrow to the bin's own value range and iterate
        float nmn = INF, nmx = -INF;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            if (x[t] >= mn && x[t] <= mx) {
                int b = (int)((x[t] - mn) * scale);
                b = b > SEL_BINS - 1 ? SEL_BINS - 1 : b;
                if (b == binsel) { nmn = fminf(nmn, x[t]); nmx = fmaxf(nmx, x[t]); }
            }
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
            if (x[t] > result) nx = fminf(nx, x[t]);
        }
        res.cnt_le = wave_sum_i(cle);
        res.next = wave_min(nx);
    }
    return res;
}

// largest f32 x with sqrt(x) <= eps (inclusive) or sqrt(x) < eps (exclusive); -1 if none
__device__ __forceinline__ float d2_threshold(float eps, int inclusive)
{
    if (!(eps >= 0.0f)) return -1.0f;
    if (eps == __builtin_inff()) return eps;
    float c = __fmul_rn(eps, eps);
    if (inclusive) {
        for (int it = 0; it < 8 && __builtin_sqrtf(c) > eps; ++it)
            c = __int_as_float(__float_as_int(c) - 1);
        if (__builtin_sqrtf(c) > eps) return -1.0f;
        for (int it = 0; it < 8; ++it) {
            float up = __int_as_float(__float_as_int(c) + 1);
            if (__builtin_sqrtf(up) <= eps) c = up; else break;
        }
        return c;
    } else {
        if (eps == 0.0f) return -1.0f;
        for (int it = 0; it < 8 && !(__builtin_sqrtf(c) < eps); ++it) {
            if (c == 0.0f) return -1.0f;
            c = __int_as_float(__float_as_int(c) - 1);
        }
        if (!(__builtin_sqrtf(c) < eps)) return -1.0f;
        for (int it = 0; it < 8; ++it) {
            float up = __int_as_float(__float_as_int(c) + 1);
            if (__builtin_sqrtf(up) < eps) c = up; else break;
        }
        return c;
    }
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

__global__ void sqrt_probe_kernel(const float *in, float *out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_sqrtf(in[i]);
}

}  // namespace acx
