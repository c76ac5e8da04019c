// EarlyFusion cross-similarity GEMMs, round 6 experiment: WAVE-SPECIALISED persistent workgroups -- the epilogue of a tile runs on waves of
// its own while the matrix waves are already multiplying the next tile.  (ACX_EF_WS=1; fp16 arithmetic; profiles/r06_ef.md says what it measured.)
//
// Reference: the X.dot(Y.T) of get_csm / get_csm_cosine (acoss/utils/cross_recurrence.py:30-73), as ef_gemm_rect_bf16x3_kernel<CH, 1> computes
// them: the same 32-k chunks, the same three products per cell in the same order, the same epilogue arithmetic per element -- bit-identical
// matrices.  Why: in every kernel so far a tile's epilogue (~14 k cycles: two load latencies, ~1000 vector instructions per wave, LDS turns,
// stores) runs on the waves that own the accumulators, and the matrix pipe idles meanwhile (0.59 busy at best); a second accumulator set to
// overlap it with the next tile's MFMAs does not fit 256 registers at 64 x 64 cells per wave.  Here:
//   * a workgroup is 12 waves (3 per SIMD, <= 168 registers): 8 MATRIX waves of 32 x 64 cells each (a 128 x 128 tile; 32 accumulator
//     registers) and 4 EPILOGUE waves;
//   * the LDS holds three operand buffers (3 x 32 KB; LDS-DMA staging as in ef_gemm_dma_kernels.hpp, four 1 KB pieces per matrix wave and chunk)
//     and ONE parked output tile (128 x 128 f32 = 64 KB): 160 KB;
//   * at the end of a tile the matrix waves drop their raw accumulators into the parking tile (8 ds_write_b128 per wave, rows XOR-swizzled)
//     and start the next tile at once; the epilogue waves turn the parked tile into distances -- norms, scales, sqrt, full-line stores, the
//     rims element by element -- one 16 x 32 unit per chunk of the matrix waves' k loop, in step with its barriers;
//   * two barriers per tile boundary (the parking tile is free / the parking tile is full and the next tile's first chunk has landed).
// Tiles are dealt statically (tile = workgroup + k x workgroups): both kinds of waves walk the same sequence without talking.
#pragma once
#include "ef_kernels.hpp"

namespace acx {

constexpr int EFW_ROWS = 128, EFW_COLS = 128, EFW_MW = 8, EFW_EW = 4, EFW_THREADS = 64 * (EFW_MW + EFW_EW);
constexpr int EFW_A = 2 * EFW_ROWS * EFB_LP, EFW_B = 2 * EFW_COLS * EFB_LP;      // fp16 elements of one operand buffer (two terms)
constexpr int EFW_PARK_PITCH = 128;                                              // floats per parked row (32 pieces of 16 bytes, piece ^ (row & 15))
constexpr int EFW_LDS_BYTES = 2 * 3 * (EFW_A + EFW_B) + EFW_ROWS * EFW_PARK_PITCH * 4;
static_assert(EFW_LDS_BYTES == 163840, "three operand buffers and the parked tile fill the LDS");

template <int CH>
__global__ __launch_bounds__(EFW_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void ef_gemm_rect_ws_kernel(
    const unsigned short *__restrict__ split0, const unsigned short *__restrict__ split1, const float *__restrict__ nrm0,
    const float *__restrict__ nrm1, const EfPair *__restrict__ pd, const EfSegRect *__restrict__ rects,
    const EfSegWg *__restrict__ wgs, const EfSegGroup *__restrict__ rowg, const EfSegGroup *__restrict__ colg, const int32_t *__restrict__ pairtab,
    float *__restrict__ scratch, int Kp0, int Kp1, const float *__restrict__ inv0, const float *__restrict__ inv1,
    int ntiles, int nfeat)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short efr_lds[];
    unsigned short *As = efr_lds;                    // [buffer][term][row][32 k]
    unsigned short *Bs = efr_lds + 3 * EFW_A;
    float *Park = reinterpret_cast<float *>(efr_lds + 3 * (EFW_A + EFW_B));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NT = 2;
    const int total = ntiles * nfeat;                // feature-major: all mfcc tiles, then all ssm tiles
    const int stride = (int)gridDim.x;
    if ((int)blockIdx.x >= total) return;
    auto feat_of = [&](int t) { return CH ? 2 : (t >= ntiles ? 1 : 0); };
    auto kp_of = [&](int s) { return (CH || s == 0) ? Kp0 : Kp1; };
    auto pool_of = [&](int s) { return (CH || s == 0) ? split0 : split1; };
    auto inc = [](int b) { return b == 2 ? 0 : b + 1; };
    auto park_at = [&](int row, int piece) { return Park + row * EFW_PARK_PITCH + 4 * (piece ^ (row & 15)); };
    auto bar = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
    };

    if (wave < EFW_MW) {
        // =============================================== matrix waves ===============================================
        const int wr = wave >> 1, wc = wave & 1;      // rows 32 wr .. + 31, columns 64 wc .. + 63 of the tile
        const int lr = lane & 15, lk = lane >> 4;
        constexpr int NA = 2, NB = 4;
        // staging: this wave's lanes land on the 16 rows of block `wave` of A and of B; lane 4 r + pos fetches logical piece pos ^ swz(row)
        const int srow = 16 * wave + (lane >> 2);
        const int sp = (lane & 3) ^ ((0x78 >> (2 * ((srow >> 2) & 3))) & 3);
        const int sr = lane >> 2;
        const unsigned short *ap = nullptr, *bp = nullptr;
        int spa = 0, pieces = 1;
        // the staging pointers of tile t (wave-uniform scalar chain: tile -> rectangle -> the two staging groups -> (CH) the roll)
        struct Stg { EfSegWg W; EfSegRect R; int rowA, vA, slotA, rowB, vB, p; bool ok; };
        auto stg1 = [&](int t, Stg &Q) { Q.ok = t < total; Q.W = wgs[Q.ok ? (t >= ntiles ? t - ntiles : t) : 0]; };
        auto stg2 = [&](Stg &Q) { Q.R = rects[Q.W.rect]; };
        auto stg3 = [&](Stg &Q) {
            const bool inA = 8 * Q.W.ty + wave < Q.R.ng, inB = wave < Q.W.pad;
            const EfSegGroup gA = rowg[Q.R.g0 + (inA ? 8 * Q.W.ty + wave : 0)];
            const EfSegGroup gB = colg[Q.R.h0 + Q.W.tx + (inB ? wave : 0)];
            Q.rowA = (int)gA.poolrow; Q.vA = inA ? gA.valid : 0; Q.slotA = gA.slot;
            Q.rowB = (int)gB.poolrow; Q.vB = inB ? gB.valid : 0;
            Q.p = CH ? colg[Q.R.h0 + Q.W.tx].slot : 0;
        };
        auto stg4 = [&](Stg &Q) { if (CH) Q.p = pairtab[Q.R.ptab0 + Q.slotA * Q.R.ncols + Q.p]; };
        auto stg5 = [&](Stg &Q) { if (CH) Q.p = Q.p >= 0 ? pd[Q.p].oti : 0; };
        auto stg6 = [&](int t, const Stg &Q) {
            const int s_ = feat_of(t);
            const int Kp = kp_of(s_);
            const unsigned short *S = pool_of(s_);
            pieces = Kp / 8;
            ap = S; bp = S + sp * 8; spa = sp;
            if (sr < Q.vA) {
                ap = S + ((int64_t)Q.rowA + sr) * NT * Kp;
                if (CH) { int r = (sp - (pieces / 12) * Q.p) % pieces; spa = r < 0 ? r + pieces : r; }
            }
            if (sr < Q.vB) bp = S + ((int64_t)Q.rowB + sr) * NT * Kp + sp * 8;
            if (!CH) ap += sp * 8;
        };
        typedef __attribute__((address_space(3))) void lds_void_t;
        typedef __attribute__((address_space(1))) const void gbl_void_t;
        // piece j of a chunk: 0 / 1 the two terms of A block `wave`, 2 / 3 of B block `wave`
        auto dma_piece = [&](int buf, auto j_tag) {
            constexpr int j = decltype(j_tag)::value;
            constexpr int t = j & 1, which = j >> 1;
            const unsigned short *src;
            if (which == 1) src = bp;
            else if (!CH) src = ap;
            else src = ap + (32 * NT) * (spa >> 2) + 8 * (spa & 3);
            src += t * EFB_BK;
            unsigned short *dst = which == 1 ? Bs + buf * EFW_B + (t * EFW_COLS + 16 * wave) * EFB_LP
                                             : As + buf * EFW_A + (t * EFW_ROWS + 16 * wave) * EFB_LP;
            __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)dst, 16, 0, 0);
        };
        auto dma_advance = [&]() {
            bp += NT * EFB_BK;
            if (!CH) ap += NT * EFB_BK;
            else { spa += 4; spa = spa >= pieces ? spa - pieces : spa; }
        };
        auto for4 = [&](auto &&f) {
            f(std::integral_constant<int, 0>()); f(std::integral_constant<int, 1>()); f(std::integral_constant<int, 2>()); f(std::integral_constant<int, 3>());
        };
        const int lks = lk ^ ((0x78 >> (2 * ((lr >> 2) & 3))) & 3);
        const unsigned short *aop = As + (32 * wr + lr) * EFB_LP + 8 * lks;
        const unsigned short *bop = Bs + (64 * wc + lr) * EFB_LP + 8 * lks;
        f32x4 acc[NA][NB];
        bf16x8 pa0[NA], pa1[NA], pb0, pb1;
        auto prefetch = [&](int buf) {
            const unsigned short *a_ = aop + buf * EFW_A, *b_ = bop + buf * EFW_B;
            pb1 = *reinterpret_cast<const bf16x8 *>(b_ + EFW_COLS * EFB_LP);
#pragma unroll
            for (int a = 0; a < NA; ++a) pa0[a] = *reinterpret_cast<const bf16x8 *>(a_ + (16 * a) * EFB_LP);
            pb0 = *reinterpret_cast<const bf16x8 *>(b_);
#pragma unroll
            for (int a = 0; a < NA; ++a) pa1[a] = *reinterpret_cast<const bf16x8 *>(a_ + (EFW_ROWS + 16 * a) * EFB_LP);
        };
        // one chunk out of buffer `cur` (products x1 y2, x2 y1, x1 y1 per cell: ef_gemm_rect_bf16x3_kernel<CH, 1>'s order)
        auto chunk = [&](int cur, int nxt, int nn, auto st_tag, auto ld_tag) {
            constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
            constexpr int NG = 3;
            constexpr int TA[3] = {0, 1, 0}, TB[3] = {1, 0, 0};
            constexpr int LASTP = 8;
            const unsigned short *b_ = bop + cur * EFW_B;
            bf16x8 av[NA][2], bv[2][2];
            auto rdb = [&](int e, int b, int q) { bv[e][q] = *reinterpret_cast<const bf16x8 *>(b_ + (q * EFW_COLS + 16 * b) * EFB_LP); };
#pragma unroll
            for (int a = 0; a < NA; ++a) { av[a][0] = pa0[a]; av[a][1] = pa1[a]; }
            bv[0][1] = pb1; bv[0][0] = pb0;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
#pragma unroll
                    for (int a = 0; a < NA; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bv[b & 1][TB[g]]), __builtin_bit_cast(f16x8, av[a][TA[g]]), acc[a][b], 0, 0, 0);
                    if (g == 1 && b + 1 < NB) { rdb((b + 1) & 1, b + 1, 1); rdb((b + 1) & 1, b + 1, 0); }
                    const int slot = NG * b + g;
                    if (LD && slot >= 1 && slot <= 4) {
                        switch (slot - 1) {
                        case 0: dma_piece(nn, std::integral_constant<int, 0>()); break;
                        case 1: dma_piece(nn, std::integral_constant<int, 1>()); break;
                        case 2: dma_piece(nn, std::integral_constant<int, 2>()); break;
                        default: dma_piece(nn, std::integral_constant<int, 3>()); break;
                        }
                    }
                    if (ST && slot == LASTP) {
                        if (LD) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        bar();
                        prefetch(nxt);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (LD) dma_advance();
        };

        int t = (int)blockIdx.x;
        Stg Q;
        stg1(t, Q); stg2(Q); stg3(Q); stg4(Q); stg5(Q);
        stg6(t, Q);
        int base = 0;
        for4([&](auto j_tag) { dma_piece(0, j_tag); });
        dma_advance();
        for (bool first = true;; first = false) {
            const int s = feat_of(t);
            const int nk = kp_of(s) / EFB_BK;
            const int t_next = t + stride;
            if (first) {
                if (nk > 1) {
                    for4([&](auto j_tag) { dma_piece(1, j_tag); });
                    dma_advance();
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bar();                                  // (the first tile's chunk 0 has landed; the epilogue waves mirror this barrier)
            }
            float zero_ = 0.0f;
            asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{zero_, zero_, zero_, zero_};
            stg1(t_next, Q);
            const bool hooked = nk >= 7;
            int cur = base;
            prefetch(cur);
            int kc = 0;
            for (; kc + 2 < nk; ++kc) {
                chunk(cur, inc(cur), inc(inc(cur)), std::true_type(), std::true_type());
                cur = inc(cur);
                if (hooked && kc < 4) {
                    if (kc == 0) stg2(Q);
                    else if (kc == 1) stg3(Q);
                    else if (kc == 2) stg4(Q);
                    else stg5(Q);
                }
            }
            if (kc + 1 < nk) {
                chunk(cur, inc(cur), inc(inc(cur)), std::true_type(), std::false_type());
                cur = inc(cur);
                ++kc;
            }
            chunk(cur, inc(cur), inc(inc(cur)), std::false_type(), std::false_type());
            if (!hooked) { stg2(Q); stg3(Q); stg4(Q); stg5(Q); }
            // the next tile's chunks 0 and 1 into the two buffers the last two barriers freed
            const int nbase = inc(cur);
            const int nk_next = kp_of(feat_of(t_next)) / EFB_BK;
            stg6(t_next, Q);
            if (Q.ok) {
                for4([&](auto j_tag) { dma_piece(nbase, j_tag); });
                dma_advance();
                if (nk_next > 1) {
                    for4([&](auto j_tag) { dma_piece(inc(nbase), j_tag); });
                    dma_advance();
                }
            }
            bar();                                      // X: the epilogue waves are done with the parked tile
            // drop the accumulators: sub-tile (a, b) -> rows 32 wr + 16 a + lr, pieces 16 wc + 4 b + lk of the parked tile
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    *reinterpret_cast<f32x4 *>(park_at(32 * wr + 16 * a + lr, 16 * wc + 4 * b + lk)) = acc[a][b];
            if (Q.ok) {
                if (nk_next > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            bar();                                      // Y: the parked tile is complete, the next tile's chunk 0 has landed
            if (!Q.ok) break;
            t = t_next;
            base = nbase;
        }
        return;
    }

    // =============================================== epilogue waves ===============================================
    const int e = wave - EFW_MW;                      // rows 32 e .. 32 e + 31 of the parked tile: row groups 2 e, 2 e + 1 of the tile
    const int lr = lane & 15, lk = lane >> 4;
    struct Tile {                                     // (packed as in ef_gemm_persist_kernels.hpp)
        int rowA[2], pkA[2], rowB[8], pkB[8];
        unsigned pp[8];                               // pair index + 1 of sub-tile 8 ua + cb, 16 bits each
        int s;
        __device__ __forceinline__ int validA(int a) const { return pkA[a] & 31; }
        __device__ __forceinline__ int validB(int b) const { return pkB[b] & 31; }
        __device__ __forceinline__ int localA(int a) const { return pkA[a] >> 5; }
        __device__ __forceinline__ int localB(int b) const { return pkB[b] >> 5; }
        __device__ __forceinline__ int pidx(int a, int b) const { return (int)((pp[(8 * a + b) >> 1] >> (16 * ((8 * a + b) & 1))) & 0xffffu) - 1; }
    };
    // a tile's description in four stages of wave-uniform scalar loads (tile and rectangle; row groups; column groups; pair table): one stage
    // per slice of the chunk loop, so that no slice waits out the whole dependent chain
    struct Res { EfSegWg W; EfSegRect R; int slotA[2], slotB[8]; };
    auto res1 = [&](int t, Res &Q) { Q.W = wgs[t >= ntiles ? t - ntiles : t]; };
    auto res2 = [&](Res &Q) { Q.R = rects[Q.W.rect]; };
    auto res3 = [&](Res &Q, Tile &T) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const bool in = 8 * Q.W.ty + 2 * e + a < Q.R.ng;
            const EfSegGroup g = rowg[Q.R.g0 + (in ? 8 * Q.W.ty + 2 * e + a : 0)];
            T.rowA[a] = (int)g.poolrow; T.pkA[a] = (g.local0 << 5) | (in ? g.valid : 0); Q.slotA[a] = g.slot;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool in = b < Q.W.pad;
            const EfSegGroup g = colg[Q.R.h0 + Q.W.tx + (in ? b : 0)];
            T.rowB[b] = (int)g.poolrow; T.pkB[b] = (g.local0 << 5) | (in ? g.valid : 0); Q.slotB[b] = g.slot;
        }
    };
    auto res4 = [&](int t, Res &Q, Tile &T) {
#pragma unroll
        for (int k = 0; k < 8; ++k) T.pp[k] = 0u;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                int p = pairtab[Q.R.ptab0 + Q.slotA[a] * Q.R.ncols + Q.slotB[b]];
                p = (T.validA(a) > 0 && T.validB(b) > 0) ? p : -1;
                p = __builtin_amdgcn_readfirstlane(p);
                T.pp[(8 * a + b) >> 1] |= (unsigned)(p + 1) << (16 * ((8 * a + b) & 1));
            }
        T.s = feat_of(t);
    };
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    // unit u of the parked tile: row group ua = u / 4, column groups 2 (u % 4) and + 1 -- the one-tile kernel's `wide` / `narrow` paths
    auto unit_c = [&](const Tile &T, auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
        constexpr int ua = u >> 2, b0 = 2 * (u & 3);
        const int s = T.s;
        const float *nrm = s == 0 ? nrm0 : nrm1;
        const float *inv = (CH || s == 0) ? inv0 : inv1;
        const int p0 = T.pidx(ua, b0), p1 = T.pidx(ua, b0 + 1);
        if (p0 < 0 && p1 < 0) return;
        int64_t cb[2];
        int pc[2], ctn[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = q ? p1 : p0;
            const EfPair *P = pd + (p < 0 ? 0 : p);
            pc[q] = P->pitchC;
            cb[q] = P->offC + (int64_t)s * P->M * pc[q] + (int64_t)T.localA(ua) * pc[q] + T.localB(b0 + q);
            ctn[q] = P->ctN;
        }
        const int prow0 = 32 * e + 16 * ua;           // first parked row of the unit
        auto value1 = [&](float dot0, float nxv, float nyv, float sxv, float syv) {
            const float dot = (dot0 * sxv) * syv;
            if (CH) return 1.0f - dot;
            float tq = (nxv + nyv) - 2.0f * dot;
            if (tq < 0.0f) tq = 0.0f;
            return ef_sqrt_nonneg(tq);
        };
        const bool wide = p0 >= 0 && p0 == p1 && T.validA(ua) == 16 && T.validB(b0) == 16 && T.validB(b0 + 1) == 16 &&
                          T.localB(b0 + 1) == T.localB(b0) + 16 && !ctn[0];
        if (wide) {
            // rows tr and 8 + tr, columns 4 pc4 .. + 3 of the unit's 32: full 128-byte lines
            const int tr = lane >> 3, pc4 = lane & 7;
            const int bq = pc4 >> 2, cl = 4 * (pc4 & 3);                   // the lane's column group of the two, its first column inside it
            const int rowB = bq ? T.rowB[b0 + 1] : T.rowB[b0];
            float nxv[2] = {0.f, 0.f};
            f32x4 nyv = {0.f, 0.f, 0.f, 0.f};
            if (!CH) {
                nxv[0] = nrm[T.rowA[ua] + tr]; nxv[1] = nrm[T.rowA[ua] + 8 + tr];
                nyv = *reinterpret_cast<const f32x4u *>(nrm + rowB + cl);
            }
            const float sxv[2] = {inv[T.rowA[ua] + tr], inv[T.rowA[ua] + 8 + tr]};
            const f32x4 syv = *reinterpret_cast<const f32x4u *>(inv + rowB + cl);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(park_at(prow0 + 8 * h + tr, 4 * b0 + pc4));
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = value1(w[r], nxv[h], nyv[r], sxv[h], syv[r]);
                float *cr = scratch + cb[0] + (int64_t)(8 * h + tr) * pc[0] + 4 * pc4;
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(cr));
            }
            return;
        }
        // rims and track ends: one sub-tile at a time, row lr, columns 4 lk .. + 3 (the matrix waves' own layout)
        const int il = lr, jl = 4 * lk;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = q ? p1 : p0;
            if (p < 0) continue;
            const int b = b0 + q;
            const f32x4 w = *reinterpret_cast<const f32x4 *>(park_at(prow0 + il, 4 * b + lk));
            float nxv = 0.f;
            f32x4 nyv = {0.f, 0.f, 0.f, 0.f};
            if (!CH) { nxv = nrm[T.rowA[ua] + il]; nyv = *reinterpret_cast<const f32x4u *>(nrm + T.rowB[b] + jl); }
            const float sxv = inv[T.rowA[ua] + il];
            const f32x4 syv = *reinterpret_cast<const f32x4u *>(inv + T.rowB[b] + jl);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = value1(w[r], nxv, nyv[r], sxv, syv[r]);
            float *cr = scratch + cb[q] + (int64_t)il * pc[q] + jl;
            if (T.validA(ua) == 16 && T.validB(b) == 16) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(cr));
            else if (il < T.validA(ua)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (jl + r < T.validB(b)) cr[r] = v[r];
            }
            if (ctn[q] && il < T.validA(ua)) {
                const EfPair P = pd[p];
                float *ct = scratch + ef_ct_off(P, s) + (size_t)(T.localB(b) + jl) * P.pitchT + T.localA(ua) + il;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (jl + r < T.validB(b)) ct[(size_t)r * P.pitchT] = v[r];
            }
        }
    };
    // (the unit index as a compile-time constant: a runtime index into the tile's arrays sends them to scratch memory)
    auto unit = [&](const Tile &T, int u) {
#ifdef ACX_EFW_ABL_NOEPI       /* ablation (WRONG matrices): the epilogue waves only keep step with the barriers -- what the matrix waves do alone */
        (void)T; (void)u; return;
#endif
        switch (u) {
        case 0: unit_c(T, std::integral_constant<int, 0>()); break;
        case 1: unit_c(T, std::integral_constant<int, 1>()); break;
        case 2: unit_c(T, std::integral_constant<int, 2>()); break;
        case 3: unit_c(T, std::integral_constant<int, 3>()); break;
        case 4: unit_c(T, std::integral_constant<int, 4>()); break;
        case 5: unit_c(T, std::integral_constant<int, 5>()); break;
        case 6: unit_c(T, std::integral_constant<int, 6>()); break;
        default: unit_c(T, std::integral_constant<int, 7>()); break;
        }
    };
    {
        Tile T, Tn;
        Res Q;
        int t = (int)blockIdx.x;
        bool have = false;                            // a parked tile (described by T) is waiting
        res1(t, Q); res2(Q); res3(Q, Tn); res4(t, Q, Tn);
        for (bool first = true;; first = false) {
            const int nk = kp_of(feat_of(t)) / EFB_BK;
            const int t_next = t + stride;
            if (first) bar();
            // In step with the matrix waves' k loop of tile t -- one barrier per chunk but the last.  Before each: a unit of the parked tile
            // (the previous one), then a stage of THIS tile's description (it is parked at this iteration's end).
            int u = 0, st = first ? 4 : 0;
            for (int kc = 0; kc + 1 < nk; ++kc) {
                if (have && u < 8) { unit(T, u); ++u; }
                else if (st < 4) {
                    if (st == 0) res1(t, Q);
                    else if (st == 1) res2(Q);
                    else if (st == 2) res3(Q, Tn);
                    else res4(t, Q, Tn);
                    ++st;
                }
                bar();
            }
            if (have)
                for (; u < 8; ++u) unit(T, u);
            for (; st < 4; ++st) {
                if (st == 0) res1(t, Q);
                else if (st == 1) res2(Q);
                else if (st == 2) res3(Q, Tn);
                else res4(t, Q, Tn);
            }
            bar();                                    // X: the parking tile is free
            T = Tn;
            have = true;
            bar();                                    // Y: tile t is parked
            if (t_next >= total) break;
            t = t_next;
        }
        for (int u = 0; u < 8; ++u) unit(T, u);       // the last tile
    }
}

}  // namespace acx
