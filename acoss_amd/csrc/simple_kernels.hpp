// SiMPle device kernels for gfx950 (reference: acoss/algorithms/simple_silva.py:45-54, 68-126).
//
// Per ORDERED pair (i, j) (the reference runs permutations, coverid.py:135):
//   OTI      shift = last argmax_s <sum_t A, roll(sum_t B, s)>       simple_silva.py:45-54
//   profile  MP[a] = min_b  sum_{c<12, k<L} (A[c,a+k] - B'[c,b+k])^2  simple_silva.py:68-116
//            evaluated like the reference as |a|^2 + |b|^2 - 2 <a, b>, all in f64
//   score    D[i, j] = -median(MP)                                    simple_silva.py:118,125
//
// One WAVE per ordered pair, no workgroup barriers, the distance matrix never exists:
//   * lane = row a of the profile (subsequence start in A), rows in groups of 64; the wave steps
//     through the columns b.  Everything that belongs to the column -- the two frames of B that
//     enter and leave the window, |b|^2 -- is wave-uniform and arrives through the SCALAR cache;
//     the lane's two frames of A (entering / leaving, already rotated by the OTI shift) and |a|^2
//     sit in registers for the whole sweep: the 12-dim frame products are v_fma_f64 with an SGPR
//     operand, no LDS, no MFMA operand staging;
//   * <a, b> slides down the diagonal exactly as the reference's STOMP update does
//     (simple_silva.py:107-110): dot[a][b] = dot[a-1][b-1] - <A[a-1], B'[b-1]> + <A[a+L-1], B'[b+L-1]>;
//     dot[a-1][b-1] is the neighbour lane's value of the previous step (one 64-bit lane shift),
//     for the first row of a group the value the previous row group's last lane left in LDS; the leaving
//     product is the entering product of the lane L below, L steps ago (register ring + lane permute: one
//     12-term product per cell instead of two); row 0 (and column 0) are evaluated in full up front;
//   * the row minimum is a running v_min_f64 in the lane's own register -- no atomics, no reduction;
//   * the median of the profile is an exact binary search on order-preserving 64-bit keys, counted
//     with ballots.
// LDS per pair: 8 (nb + na + row groups) bytes -- tracks of thousands of pooled frames fit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acx {

constexpr int SIMPLE_MAXN = 6000;    // pooled frames per track (LDS per wave: 8 (nb + na + row groups) + 64 bytes = 97 KB at 6000)
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

// |x_t|^2 + ... + |x_{t+L-1}|^2 for every window start t of every track, once per (pool, L):
// wn[toff[track] + t], t <= n - L (np.sum(np.square(seq[:, t:t+L])), simple_silva.py:85-89).
static __global__ void simple_winnorm_kernel(const double *__restrict__ pool, const int64_t *__restrict__ toff,
                                             double *__restrict__ wn, int L)
{
    const int track = blockIdx.x;
    const int64_t t0 = toff[track];
    const int n = (int)(toff[track + 1] - t0);
    for (int t = blockIdx.y * 256 + threadIdx.x; t <= n - L; t += 256 * gridDim.y) {
        double acc = 0.0;
        for (int k = 0; k < L; ++k)
            for (int c = 0; c < 12; ++c) {
                const double v = pool[(t0 + t + k) * 12 + c];
                acc += v * v;
            }
        wn[t0 + t] = acc;
    }
}

// k-th smallest (0-based) of the m keys in LDS: binary search on the key value, every probe counted
// with ballots.  Exact; ties are harmless (the answer is a value, not a position).
__device__ __forceinline__ unsigned long long simple_select_key(const unsigned long long *keys, int m, int k, int lane)
{
    unsigned long long lo = 0ull, hi = 0xffffffffffffffffull;      // answer in [lo, hi]
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        int tot = 0;
        for (int i0 = 0; i0 < m; i0 += 64) {                       // wave-uniform trip count
            const int i = i0 + lane;
            const bool le = i < m && keys[i] <= mid;
            tot += __popcll(__ballot(le));
        }
        if (tot >= k + 1) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// pool: (sum n_t, 12) f64 time-major; prof: (n_tracks, 12) f64 = sum over time of every bin;
// wn: window norms (simple_winnorm_kernel, same L).  One WAVE per ordered pair, SIMPLE_WPB independent
// waves per workgroup (no barrier anywhere): the host hands the pairs over sorted by their second
// track, so the waves of a workgroup -- and of its neighbours on the CU -- walk the SAME frames of B at
// about the same time and share their scalar-cache misses.
constexpr int SIMPLE_WPB = 4;
// Waves per SIMD the kernel is compiled for: seven (72 registers) up to the reference's own subsequence length, L = 10 -- 71 registers,
// nothing spilled: 14.46 -> 13.81 ms per 262 k ordered pairs against the six waves of the compiler's own choice (74 registers); eight
// spill (21.4 ms), and from L = 12 on the ring (2 L registers) spills at seven.  -DACX_SIMPLE_WAVES=N overrides (A/B builds).
#ifdef ACX_SIMPLE_WAVES
constexpr int simple_waves(int) { return ACX_SIMPLE_WAVES; }
#else
constexpr int simple_waves(int l) { return l <= 10 ? 7 : 1; }
#endif
template <int L>
__global__ __launch_bounds__(64 * SIMPLE_WPB, simple_waves(L)) void simple_kernel(const double *__restrict__ pool,
                                                                 const int64_t *__restrict__ toff,
                                                                 const double *__restrict__ prof,
                                                                 const double *__restrict__ wn,
                                                                 const int32_t *__restrict__ pairs,
                                                                 double *__restrict__ out, int do_oti, int npairs,
                                                                 int smem_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pidx = blockIdx.x * (int)(blockDim.x >> 6) + wave;       // (long tracks: fewer waves per workgroup, LDS)
    if (pidx >= npairs) return;                                  // wave-uniform
    unsigned char *smem_raw = smem_all + (size_t)wave * smem_per_wave;
    const int ti = pairs[2 * pidx], tj = pairs[2 * pidx + 1];
    const int64_t oa = toff[ti], ob = toff[tj];
    const int na = (int)(toff[ti + 1] - oa), nb = (int)(toff[tj + 1] - ob);
    const int ma = na - L + 1, mb = nb - L + 1;      // profile length, columns
    // E: the dot products a row group hands to the next one (its last lane's value of every column; before group 0:
    // row 0 in full).  ONE array serves all groups: group g reads column c of its predecessor at slot c + eoff and
    // stores its own column c one slot lower, at c + eoff - 1 -- the slot it has just read (column c - 1's) --, and
    // the next group reads with eoff - 1.  (Two alternating arrays cost a third more LDS per wave, and LDS is what
    // limits this kernel to 6 waves per SIMD; with one array 8 fit.)
    constexpr int STRIDE_ = 64 - L;
    const int ngroups_ = (ma + STRIDE_ - 1) / STRIDE_;
    double *E = reinterpret_cast<double *>(smem_raw);            // mb + ngroups + 1 slots
    unsigned long long *mp = reinterpret_cast<unsigned long long *>(E + mb + ngroups_ + 1);   // ma keys of the profile

    // ---- OTI (simple_silva.py:45-54): v[s] = <pa, roll(pb, s)>; np.argsort(v)[-1]  (wave-uniform)
    int shift = 0;
    if (do_oti) {
        const double *pa = prof + (size_t)ti * 12, *pb = prof + (size_t)tj * 12;
        double bestv = 0.0;
        for (int s = 0; s < 12; ++s) {
            double acc = 0.0;
            for (int c = 0; c < 12; ++c) acc += pa[c] * pb[(c - s + 12) % 12];
            if (s == 0 || acc >= bestv) { bestv = acc; shift = s; }     // ties: highest index (stable sort order)
        }
    }
    // B' = roll(B, shift): B'[u][c] = B[u][(c - shift) mod 12], so <A[t], B'[u]> = sum_c A[t][(c + shift) mod 12] B[u][c]:
    // the lane's frames of A are loaded rotated, the frames of B are used as stored (scalar loads)
    const double *ga = pool + oa * 12, *gb = pool + ob * 12;
    const double *wa = wn + oa, *wb = wn + ob;
    auto load_a = [&](int t, double (&v)[12]) {
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            int cs = c + shift; if (cs >= 12) cs -= 12;
            v[c] = ga[(size_t)t * 12 + cs];
        }
    };
    auto dot12 = [&](const double (&a)[12], const double *brow) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 12; ++c) acc = __builtin_fma(a[c], brow[c], acc);
        return acc;
    };

    // ---- row 0 in full: top[b] = sum_k <A[k], B'[b + k]>  (lanes over b), kept in E (slots b + ngroups_) as the "row before" of the
    // first group; row 0's own profile value -- the smallest of its distances -- is taken here and the groups start at row 1 (inside
    // group 0 the row cost every step of every group a 64-bit select: "row 0 takes the stored value")
    {
        unsigned long long k0 = 0xffffffffffffffffull;
        const double a2_0 = wa[0];
        for (int b = lane; b < mb; b += 64) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < L; ++k) {
                double ak[12];
                load_a(k, ak);                                   // wave-uniform address: scalar loads
                acc += dot12(ak, gb + (size_t)(b + k) * 12);
            }
            E[b + ngroups_] = acc;
            const unsigned long long kd = f64_key((a2_0 + wb[b]) - 2.0 * acc);
            k0 = kd < k0 ? kd : k0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(k0, o);
            k0 = other < k0 ? other : k0;
        }
        if (lane == 0) mp[0] = k0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // Row groups of 64 lanes: L FEEDER lanes + (64 - L) rows.  The product that leaves the window of row a at
    // step b, <A[a - 1], B'[b - 1]>, is the product that ENTERED the window of row a - L at step b - L: lane l - L
    // computed it L steps ago.  So a lane computes ONE 12-term product per step (the entering one), hands it L
    // lanes up (one 64-bit lane permute, consumed L steps later: its latency is free) and keeps the L it has
    // received in a register ring; the leaving frame of B is never loaded again.  The first L lanes of a group
    // have nobody below them: they only feed -- they repeat the last L rows of the previous group (group 0: the
    // rows 1 - L .. 0; the frames they multiply, A[0 .. L - 1], exist).
    constexpr int STRIDE = 64 - L;
    static_assert(L >= 1 && L <= 16, "feeder lanes");
    const int ngroups = (ma - 1 + STRIDE - 1) / STRIDE;             // rows 1 .. ma - 1 (<= ngroups_, which sizes E)
    // The lanes are ROTATED: the group's first row sits in physical lane 0, its feeders in the top L lanes (`vl`, the position in the
    // group, = lane + L mod 64).  The value a row takes from the row before it -- one lane down, a DPP wave_shr:1 -- has no source in
    // lane 0, which therefore KEEPS what the destination held: the last row of the group before (E), read into it.  No select.
    const int vl = (lane + L) & 63;
    const int up_src = ((lane - L) & 63) << 2;                     // ds_bpermute address of the lane L below
    auto shift_up_L = [&](double v) {
        const long long bits = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_ds_bpermute(up_src, (int)(bits & 0xffffffffll));
        const int hi = __builtin_amdgcn_ds_bpermute(up_src, (int)(bits >> 32));
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    for (int g = 0; g < ngroups; ++g) {
        const int a = STRIDE * g + 1 + vl - L;                  // this lane's row (feeders: a row of the group before, or <= 0)
        const bool valid = vl >= L && a < ma;
        const int ac = a < 0 ? 0 : (a > ma - 1 ? ma - 1 : a);   // clamp: results of feeder / idle lanes are dropped
        int fa = a + L - 1;                                     // frame entering the window of row a
        fa = fa < 0 ? 0 : (fa > na - 1 ? na - 1 : fa);
        const int eoff = ngroups_ - g;                          // this group reads column c of its predecessor at E[c + eoff]
        const bool more = g + 1 < ngroups;
        double An[12];
        load_a(fa, An);
        const double a2 = wa[ac];
        // column 0 in full: dot[a][0] = sum_k <A[a + k], B'[k]>
        double dot = 0.0;
        {
#pragma unroll
            for (int k = 0; k < L; ++k) {
                double ak[12];
                load_a(ac + k, ak);
                dot += dot12(ak, gb + (size_t)k * 12);
            }
        }
        double mn = (a2 + wb[0]) - 2.0 * dot;
        if (more && vl == 63) E[eoff - 1] = dot;
        // ring[b % L] = the product leaving at step b.  Steps 1 .. L need the products that "entered" at steps
        // 1 - L .. 0, i.e. with the frames B'[0 .. L - 1]: computed here, before the sweep.
        double ring[L];
#pragma unroll
        for (int j = 0; j < L; ++j) ring[(1 + j) % L] = shift_up_L(dot12(An, gb + (size_t)j * 12));
        // The column's entering frame of B is wave-uniform: scalar loads into two register sets, requested TWO steps
        // ahead (the step that has just used a set refills it): a miss of the scalar cache has a step and a half to land.
        double bn[2][12];
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (1 + q < mb) {
                const double *pn = gb + (size_t)(L + q) * 12;       // step b = 1 + q multiplies frame b + L - 1
#pragma unroll
                for (int c = 0; c < 12; ++c) bn[(1 + q) & 1][c] = pn[c];
            }
        constexpr int UN = (L % 2 == 0) ? L : 2 * L;            // steps per unrolled round: ring slot and register set are static
        typedef __attribute__((address_space(3))) double lds_f64;
        typedef __attribute__((address_space(3))) void lds_void_;
        double w2[2] = {0.0, 0.0};
        static_assert(UN % 2 == 0, "window norms in pairs");
        // A round of UN steps.  FULL: every step of it exists and so does the frame two steps behind its last one -- ONE basic block, no
        // compare-and-branch per step; the last round(s) of a sweep keep the per-step tests.
        auto round = [&](int b0, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            // E's slots of this round through ONE vector register (the slot of column b0 - 1 + eoff; the steps add immediates): the
            // address came out of a scalar register into a fresh vector register for every read and every write of every step
            unsigned rbase = (unsigned)(uintptr_t)(lds_void_ *)(E + (b0 - 1 + eoff));
            asm volatile("" : "+v"(rbase));
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int b = b0 + j;                               // b % L == (1 + j) % L, b % 2 == (1 + j) % 2
                if (FULL || b < mb) {                               // wave-uniform
                    // the columns' window norms two at a time (one scalar load and one address per two steps; for L >= 2 the slot behind a
                    // track's last column lies inside the track's own range of the table)
                    if (L >= 2 ? j % 2 == 0 : true) { w2[j % 2] = wb[b]; if (L >= 2) w2[1] = wb[b + 1]; }
                    const double w = w2[j % 2];
                    // dot[a - 1][b - 1]: of the neighbour lane by two DPP wave_shr:1 moves; lane 0 -- the group's first row -- keeps the
                    // destination's old contents, the last row of the group before (E).  This value is the loop-carried dependency of the
                    // sweep; as a 64-bit __shfl_up it was two ds_bpermute_b32 round trips per step in that chain
                    const double e = *(const lds_f64 *)(rbase + 8u * j);   // E[b - 1 + eoff]: dot[a - 1][b - 1] of the group before's last lane (group 0: of row 0)
                    double prev;
                    {
                        const long long bits_ = __double_as_longlong(dot), ebits_ = __double_as_longlong(e);
                        const int lo_ = __builtin_amdgcn_update_dpp((int)(ebits_ & 0xffffffffll), (int)(bits_ & 0xffffffffll), 0x138, 0xf, 0xf, false);
                        const int hi_ = __builtin_amdgcn_update_dpp((int)(ebits_ >> 32), (int)(bits_ >> 32), 0x138, 0xf, 0xf, false);
                        prev = __longlong_as_double(((long long)hi_ << 32) | (unsigned int)lo_);
                    }
                    double gnew = 0.0;
#pragma unroll
                    for (int c = 0; c < 12; ++c) gnew = __builtin_fma(An[c], bn[(1 + j) & 1][c], gnew);
                    asm volatile("" : "+v"(gnew));                  // this register set is dead from here: it takes the frame of step b + 2
                    if (FULL || b + 2 < mb) {
                        const double *pn = gb + (size_t)(b + 1 + L) * 12;
#pragma unroll
                        for (int c = 0; c < 12; ++c) bn[(1 + j) & 1][c] = pn[c];
                    }
                    const double gold = ring[(1 + j) % L];          // entered the window of row a - L at step b - L
                    ring[(1 + j) % L] = shift_up_L(gnew);           // leaves the window of this row at step b + L
                    dot = (prev - gold) + gnew;
                    // (a2 + w) - 2 dot as ONE fused instruction: 2 dot is exact, so the fma rounds the same exact difference the
                    // subtraction rounded -- same bits, an instruction less per step
                    const double dist = __builtin_fma(-2.0, dot, a2 + w);
                    // The running minimum as ONE v_min_f64, written as the instruction: `dist < mn ? dist : mn` compiles to v_cmp_lt_f64 vcc +
                    // two v_cndmask_b32 ..., vcc next to each other, and gfx950 takes ~10 cycles for the second of two ADJACENT selects on vcc
                    // (scripts/ubench/cndmask_probe.hip, profiles/r06_cndmask_probe.txt: 17.5 cycles for the three, 4.2 for a v_min_f64);
                    // __builtin_fmin puts two canonicalising v_max_f64 in front (16.85 instead of 15.75 ms per 262 k ordered pairs, measured).
                    // Same value: neither operand is ever a NaN (the pool is checked for non-finite frames) and `x - y` never makes a -0.
                    // (Also measured, profiles/r06_simple.md: a second copy of this loop for row group 0, to drop its per-step select from
                    //  the other groups: 82 ms, the ring left the registers.)
                    asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(dist), "v"(mn));
                    if (more && vl == 63) *(lds_f64 *)(rbase + 8u * j) = dot;  // E[b + eoff - 1] (slot of column b - 1, read above)
                }
            }
        };
        int b0 = 1;
        for (; b0 + UN + 1 < mb; b0 += UN) round(b0, std::true_type());
        for (; b0 < mb; b0 += UN) round(b0, std::false_type());
        if (valid) mp[a] = f64_key(mn);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- median (np.median: mean of the two middle values for even counts)
    const int r_lo = (ma - 1) / 2, r_hi = ma / 2;
    const double vlo = key_f64(simple_select_key(mp, ma, r_lo, lane));
    const double vhi = (r_hi == r_lo) ? vlo : key_f64(simple_select_key(mp, ma, r_hi, lane));
    if (lane == 0) out[pidx] = -((r_lo == r_hi) ? vlo : (vlo + vhi) * 0.5);
}

}  // namespace acx
