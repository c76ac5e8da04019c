// EarlyFusion cross-similarity GEMMs, round 6: the rectangle kernel as a PERSISTENT workgroup with a cross-tile pipeline.
//
// Reference: the X.dot(Y.T) of get_csm / get_csm_cosine (acoss/utils/cross_recurrence.py:30-73) for every pair of a rectangle,
// as ef_gemm_rect_bf16x3_kernel<CH, F16> (ef_kernels.hpp) computes them -- same tiles, same staging layout, the SAME k loop
// (chunk by chunk the same MFMAs in the same order on the same operands: bit-identical matrices), the same epilogue arithmetic
// and stores.  What changes is what a workgroup does between two tiles.  The one-tile kernel spends 9 % of a tile's time before
// its k loop starts (workgroup launch, the dependent chain tile -> rectangle -> groups -> pair table -> rolls, the first chunk's
// trip from L2 / HBM) and 3 % waiting for its last stores to be acknowledged before the CU takes the next workgroup
// (profiles/r05_ef.md (b): 22.3 k of 232.7 k ticks); with one workgroup of 256 registers per lane and all 160 KB of LDS per CU
// nothing else runs there meanwhile.  Here ONE workgroup per CU walks tiles:
//   * tile indices come from a global counter (atomicAdd, fetched two tiles ahead by one lane and handed round through LDS at
//     the tile boundary's barrier), so the deal stays as dynamic as the hardware dispatcher's -- a static stride lost 3 % in
//     round 3;
//   * the NEXT tile's descriptor chain is resolved stage by stage behind the first chunks of the current tile's k loop (one
//     gathering load per stage into a per-lane table: no scalar registers are held for it), and its first k chunk is loaded into the staging registers (dead since the k loop's last
//     chunk) at the head of the current tile's epilogue, in front of the epilogue's stores;
//   * the stores drain under the next tile's prologue and k loop: nobody waits for them.
// A tile boundary costs two barriers (operand buffers free / first chunk visible), as a tile's start did before.
#pragma once
#include "ef_kernels.hpp"

namespace acx {

template <int CH, int F16>
__global__ __launch_bounds__(EFR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ef_gemm_rect_persist_kernel(
    const unsigned short *__restrict__ split0, const unsigned short *__restrict__ split1, const float *__restrict__ nrm0,
    const float *__restrict__ nrm1, const EfPair *__restrict__ pd, const EfSegRect *__restrict__ rects,
    const EfSegWg *__restrict__ wgs, const EfSegGroup *__restrict__ rowg, const EfSegGroup *__restrict__ colg, const int32_t *__restrict__ pairtab,
    float *__restrict__ scratch, int Kp0, int Kp1, const float *__restrict__ inv0, const float *__restrict__ inv1,
    int ntiles, int nfeat, unsigned *__restrict__ counter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short efr_lds[];
    unsigned short *As = efr_lds;                    // [buffer][term][row][32 k]
    unsigned short *Bs = efr_lds + 2 * EFR_A;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int NA = 4, NB = 4;
    constexpr int NT = F16 ? 2 : 3;
    const int total = ntiles * nfeat;                // tiles of the launch: feature-major (all mfcc tiles, then all ssm tiles)
    int t_cur = (int)blockIdx.x, t_next = (int)(blockIdx.x + gridDim.x);
    if (t_cur >= total) return;
    float *Tw = reinterpret_cast<float *>(efr_lds + 2 * (EFR_A + EFR_B)) + wave * (16 * EFR_TP);
    volatile unsigned *mailbox = reinterpret_cast<volatile unsigned *>(efr_lds + 2 * (EFR_A + EFR_B));      // (word 0 of wave 0's turning tile)

    const int srow = tid >> 2, sp = tid & 3;
    const int sg = wave, sr = srow & 15;

    // ---- a tile's description: what the one-tile kernel derives at its start, PACKED -- two tiles' worth of it is alive during the
    // k loop, and every scalar register it takes beyond the ~100 a wave has goes to a lane of a vector register, of which the k
    // loop has none to spare (a group record is 5 values, a pair index 1: 56 per tile unpacked; packed 24)
    struct Tile {
        int rowA[NA], pkA[NA];          // pool row of the group's first block; (row of the first block inside the pair's matrix) << 5 | blocks that exist (0 .. 16)
        int rowB[NB], pkB[NB];
        unsigned pp[NA * NB / 2];       // pair index + 1 of sub-tile 4 a + b (0: nothing to store), 16 bits each (a batch holds < 65 535 pairs)
        bool any;
        int s;
        __device__ __forceinline__ int validA(int a) const { return pkA[a] & 31; }
        __device__ __forceinline__ int validB(int b) const { return pkB[b] & 31; }
        __device__ __forceinline__ int localA(int a) const { return pkA[a] >> 5; }
        __device__ __forceinline__ int localB(int b) const { return pkB[b] >> 5; }
        __device__ __forceinline__ int pidx(int a, int b) const { return (int)((pp[(4 * a + b) >> 1] >> (16 * ((4 * a + b) & 1))) & 0xffffu) - 1; }
    };
    // The NEXT tile is resolved into per-lane tables, not scalar registers: every stage is ONE gathering load whose lanes fetch the
    // dwords the wave needs (lane = record x field), issued behind one chunk of the k loop and read -- v_readlane -- behind the
    // next.  Nothing of the next tile occupies a scalar register while the k loop runs (two tiles' worth of scalars plus the
    // chain's intermediates sent hundreds of spills into the loop: +3 % instead of -2 %, profiles/r06_ef.md).
    //   q1  lanes 0-3: the tile record (rect, ty, first column group, column groups)
    //   q2  lanes 0-5: its rectangle (g0, ng, h0, nh, ncols, ptab0)
    //   q3  lane 4 r + f: field f (pool row, blocks that exist, track slot, first row inside the pair's matrix) of record r --
    //       r = 0-3 the wave's row groups, 4-7 its column groups, 8 / 9 the staging row groups, 10 the staging column group,
    //       11 the tile's first column group (CH: its track is the tile's one reference track)
    //   q4  lane 4 a + b: the pair of sub-tile (a, b); lanes 16 / 17: (CH) the pairs of the two staging row groups
    //   q5  lanes 16 / 17: (CH) their rolls
    int q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0;
    bool nok = false;                                 // (wave-uniform) the next tile exists
    auto rl = [](int v, int l) { return __builtin_amdgcn_readlane(v, l); };
    auto feat_of = [&](int t) { return CH ? 2 : (t >= ntiles ? 1 : 0); };
    auto kp_of = [&](int s) { return (CH || s == 0) ? Kp0 : Kp1; };
    auto pool_of = [&](int s) { return (CH || s == 0) ? split0 : split1; };
    auto stage1 = [&](int t) {
        nok = t < total;
        const int idx = nok ? (t >= ntiles ? t - ntiles : t) : 0;
        q1 = reinterpret_cast<const int *>(wgs + idx)[lane & 3];
    };
    auto stage2 = [&]() {
        const int l6 = (lane & 7) < 6 ? (lane & 7) : 0;
        q2 = reinterpret_cast<const int *>(rects + rl(q1, 0))[l6];
    };
    // (what stage 3 and the unpacking both need of q1 / q2)
    struct Geo { int ty, tx, ncg, g0, ng, h0, gr0, gc0; bool in0, in1, inB; };
    auto geo = [&]() {
        Geo G;
        G.ty = rl(q1, 1); G.tx = rl(q1, 2); G.ncg = rl(q1, 3);
        G.g0 = rl(q2, 0); G.ng = rl(q2, 1); G.h0 = rl(q2, 2);
        G.gr0 = 16 * G.ty + NA * wr; G.gc0 = G.tx + NB * wc;
        G.in0 = 16 * G.ty + sg < G.ng; G.in1 = 16 * G.ty + 8 + sg < G.ng; G.inB = sg < G.ncg;
        return G;
    };
    auto stage3 = [&]() {
        const Geo G = geo();
        int rec = (lane >> 2) & 15;
        rec = rec > 11 ? 11 : rec;
        const int f = lane & 3;
        const int dw = f == 0 ? 0 : f + 1;             // EfSegGroup: poolrow (low dword) 0, valid 2, slot 3, local0 4
        const int k = rec & 3;
        int gi;
        if (rec < 4) gi = G.g0 + (G.gr0 + k < G.ng ? G.gr0 + k : 0);
        else if (rec < 8) gi = G.h0 + (NB * wc + k < G.ncg ? G.gc0 + k : G.tx);
        else if (rec == 8) gi = G.g0 + (G.in0 ? 16 * G.ty + sg : 0);
        else if (rec == 9) gi = G.g0 + (G.in1 ? 16 * G.ty + 8 + sg : 0);
        else if (rec == 10) gi = G.h0 + G.tx + (G.inB ? sg : 0);
        else gi = G.h0 + G.tx;
        const EfSegGroup *arr = (rec < 4 || rec == 8 || rec == 9) ? rowg : colg;
        q3 = reinterpret_cast<const int *>(arr + gi)[dw];
    };
    auto stage4 = [&]() {
        const int ncols = rl(q2, 4), ptab0 = rl(q2, 5);
        const int a = (lane >> 2) & 3, b = lane & 3;
        int la = 4 * a + 2, lb = 16 + 4 * b + 2;       // the lanes of q3 that hold the two track slots
        if (lane == 16) { la = 4 * 8 + 2; lb = 4 * 11 + 2; }
        if (lane == 17) { la = 4 * 9 + 2; lb = 4 * 11 + 2; }
        const int slotA = __builtin_amdgcn_ds_bpermute(4 * la, q3), slotB = __builtin_amdgcn_ds_bpermute(4 * lb, q3);
        const bool want = lane < 16 || (CH && lane < 18);
        q4 = pairtab[ptab0 + (want ? slotA * ncols + slotB : 0)];
    };
    auto stage5 = [&]() {
        int o = 0;
        if (CH && (lane == 16 || lane == 17) && q4 >= 0) o = pd[q4].oti;
        q5 = o;
    };
    // the resolved tile into the scalar registers of the CURRENT tile (at the tile boundary: the previous tile's are dead)
    auto unpack = [&](int t, Tile &T) {
        const Geo G = geo();
        bool any = false;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const bool in = G.gr0 + a < G.ng;
            T.rowA[a] = rl(q3, 4 * a);
            T.pkA[a] = (rl(q3, 4 * a + 3) << 5) | (in ? rl(q3, 4 * a + 1) : 0);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const bool in = NB * wc + b < G.ncg;
            T.rowB[b] = rl(q3, 16 + 4 * b);
            T.pkB[b] = (rl(q3, 16 + 4 * b + 3) << 5) | (in ? rl(q3, 16 + 4 * b + 1) : 0);
        }
#pragma unroll
        for (int k = 0; k < NA * NB / 2; ++k) T.pp[k] = 0u;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                int p = rl(q4, 4 * a + b);
                p = (T.validA(a) > 0 && T.validB(b) > 0) ? p : -1;
                any = any || p >= 0;
                T.pp[(4 * a + b) >> 1] |= (unsigned)(p + 1) << (16 * ((4 * a + b) & 1));
            }
        T.any = any;
        T.s = feat_of(t);
    };
    // the staging pointers of the resolved tile (per thread; made where the staging registers are free: the head of an epilogue)
    const unsigned short *ap0 = nullptr, *ap1 = nullptr, *bp = nullptr;
    int sp0 = 0, sp1 = 0, pieces = 1;
    auto stage6 = [&](int t) {
        const Geo G = geo();
        const int s_ = feat_of(t);
        const int Kp = kp_of(s_);
        const unsigned short *S = pool_of(s_);
        pieces = Kp / 8;
        ap0 = S; ap1 = S; bp = S + sp * 8;
        sp0 = sp; sp1 = sp;
        auto roll_of = [&](int oti) {
            int r = (sp - (pieces / 12) * oti) % pieces;     // piece - G r / 8, into [0, pieces)
            return r < 0 ? r + pieces : r;
        };
        const int v0 = G.in0 ? rl(q3, 33) : 0, v1 = G.in1 ? rl(q3, 37) : 0, vB = G.inB ? rl(q3, 41) : 0;
        if (sr < v0) { ap0 = S + ((int64_t)rl(q3, 32) + sr) * NT * Kp; if (CH) sp0 = roll_of(rl(q4, 16) >= 0 ? rl(q5, 16) : 0); }
        if (sr < v1) { ap1 = S + ((int64_t)rl(q3, 36) + sr) * NT * Kp; if (CH) sp1 = roll_of(rl(q4, 17) >= 0 ? rl(q5, 17) : 0); }
        if (sr < vB) bp = S + ((int64_t)rl(q3, 40) + sr) * NT * Kp + sp * 8;
        if (!CH) { ap0 += sp * 8; ap1 += sp * 8; }
    };

    // ---- staging (as in the one-tile kernel; the pointers are those of the tile being loaded)
    u32x4 st[9];                                      // pieces 0-2: A rows tid / 4, 3-5: A rows 128 + tid / 4, 6-8: B (one per term)
    auto gload_piece = [&](auto p_tag) {
        constexpr int p = decltype(p_tag)::value;
        if (F16 && p % 3 == 2) return;                 // (the third term does not exist)
        const unsigned short *src;
        if (p >= 6) src = bp;
        else if (!CH) src = p < 3 ? ap0 : ap1;
        else {                                         // [k / 32][term][k % 32]: piece q of the row sits at 32 NT (q / 4) + 8 (q % 4)
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
    const int skl = (sp ^ ((0x78 >> (2 * ((srow >> 2) & 3))) & 3)) * 8;
    unsigned short *as0 = As + srow * EFB_LP + skl, *bs0 = Bs + srow * EFB_LP + skl;
    auto lstore_piece = [&](int buf, auto p_tag) {
        constexpr int p = decltype(p_tag)::value;
        if (F16 && p % 3 == 2) return;
        unsigned short *dst = p < 3 ? as0 + buf * EFR_A + (p * EFR_ROWS) * EFB_LP
                            : (p < 6 ? as0 + buf * EFR_A + ((p - 3) * EFR_ROWS + 128) * EFB_LP : bs0 + buf * EFR_B + ((p - 6) * EFR_COLS) * EFB_LP);
        *reinterpret_cast<u32x4 *>(dst) = st[p];
    };
    auto for9 = [&](auto &&f) {
        f(std::integral_constant<int, 0>()); f(std::integral_constant<int, 1>()); f(std::integral_constant<int, 2>());
        f(std::integral_constant<int, 3>()); f(std::integral_constant<int, 4>()); f(std::integral_constant<int, 5>());
        f(std::integral_constant<int, 6>()); f(std::integral_constant<int, 7>()); f(std::integral_constant<int, 8>());
    };
    const int lks = lk ^ ((0x78 >> (2 * ((lr >> 2) & 3))) & 3);
    const unsigned short *aop = As + (64 * wr + lr) * EFB_LP + 8 * lks;
    const unsigned short *bop = Bs + (64 * wc + lr) * EFB_LP + 8 * lks;

    f32x4 acc[NA][NB];
    // ---- one k chunk: ef_gemm_rect_bf16x3_kernel's, instruction for instruction
    constexpr int TP = F16 ? 1 : 2;
    bf16x8 pa0[NA], pa2[NA], pb0, pb2;
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
        constexpr bool P3 = F16 && ACX_EF_F16_PRODUCTS == 3;
        constexpr int NG = F16 ? (P3 ? 3 : 4) : 6;
        constexpr int TA[6] = {F16 ? (P3 ? 0 : 1) : 0, F16 ? (P3 ? 1 : 0) : 2, P3 ? 0 : 1, 0, 1, 0},
                      TB[6] = {F16 ? 1 : 2, F16 ? (P3 ? 0 : 1) : 0, F16 ? 0 : 1, F16 ? 0 : 1, 0, 0};
        constexpr int LASTP = F16 ? (P3 ? 8 : 11) : 18;
        const unsigned short *a_ = aop + cur * EFR_A, *b_ = bop + cur * EFR_B;
        bf16x8 av[NA][3], bv[2][3];
        auto rdb = [&](int e, int b, int q) { bv[e][q] = *reinterpret_cast<const bf16x8 *>(b_ + (q * EFR_COLS + 16 * b) * EFB_LP); };
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
                    // (the builtins, not in-place inline asm: spelled out in place -- measured, profiles/r06_ef.md -- the f16x2 GEMMs take
                    //  20.3-20.4 instead of 19.8 ms; the compiler's rotation of an accumulator through the MFMAs' destinations is not
                    //  what the bf16x3 build of this kernel loses)
                    if (F16) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bv[b & 1][TB[g]]), __builtin_bit_cast(f16x8, av[a][TA[g]]), acc[a][b], 0, 0, 0);
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[b & 1][TB[g]], av[a][TA[g]], acc[a][b], 0, 0, 0);
                }
                if (g == 1 && b + 1 < NB) {
                    if (F16) { rdb((b + 1) & 1, b + 1, 1); rdb((b + 1) & 1, b + 1, 0); }
                    else { rdb((b + 1) & 1, b + 1, 2); rdb((b + 1) & 1, b + 1, 0); rdb((b + 1) & 1, b + 1, 1); }
                }
                const int slot = NG * b + g;
                if (P3 ? (slot >= 1 && slot <= 6) : (slot >= (F16 ? 1 : 2) && slot <= LASTP && (slot & 1) == (F16 ? 1 : 0))) {
                    auto piece = [&](auto p_tag) {
                        if (ST) lstore_piece(cur ^ 1, p_tag);
                        if (LD) gload_piece(p_tag);
                    };
                    const int nth = P3 ? slot - 1 : (slot - (F16 ? 1 : 2)) / 2;
                    switch (F16 ? nth + nth / 2 : nth) {
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
                    __syncthreads();
                    prefetch(cur ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (LD) gload_advance();
    };
    auto chunk_idle = [&](int cur, auto st_tag, auto ld_tag) {
        constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
        if (ST) for9([&](auto p_tag) { lstore_piece(cur ^ 1, p_tag); });
        if (LD) { for9([&](auto p_tag) { gload_piece(p_tag); }); gload_advance(); }
        if (ST) __syncthreads();
    };

    // ---- the first tile: resolved and loaded the way the one-tile kernel starts
    Tile T;
    stage1(t_cur); stage2(); stage3(); stage4(); stage5();
    unpack(t_cur, T);
    stage6(t_cur);
    for9([&](auto p_tag) { gload_piece(p_tag); });
    gload_advance();
    unsigned pending = 0;                             // (lane 0 of wave 0: the index fetched for the tile after next)

    for (bool first = true;; first = false) {
        if (!first) {
            __syncthreads();                          // every wave is out of the previous tile's k loop: the operand buffers are free
            t_next = (int)__builtin_amdgcn_readfirstlane(*mailbox);
        }
        const int s = T.s;
        const int Kp = kp_of(s);
        const int nk = Kp / EFB_BK;
        // chunk 0 (in the staging registers) into buffer 0, chunk 1 into the registers
        for9([&](auto p_tag) { lstore_piece(0, p_tag); });
        if (nk > 1) { for9([&](auto p_tag) { gload_piece(p_tag); }); gload_advance(); }
        __syncthreads();
        float zero_ = 0.0f;
        if (F16) asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));     // (a register, not the inline constant: ef_gemm_rect_bf16x3_kernel)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{zero_, zero_, zero_, zero_};
        // The next tile's descriptor chain -- five dependent gathering loads -- rides on the first chunks of this tile's k loop, one
        // stage behind each of the chunks 0 - 3 (a stage's load has a whole chunk to arrive).  Tiles of fewer than seven chunks
        // resolve it behind their k loop.
        stage1(t_next);
        const bool peeled = nk >= 7;                  // (the steady loop runs chunks 0 .. nk - 3: at least the four that carry a stage)
        auto sweep = [&](auto &&chunk) {
            int kc = 0;
            for (; kc + 2 < nk; ++kc) {
                chunk(kc & 1, std::true_type(), std::true_type());
                if (peeled && kc < 4) {               // (wave-uniform; four scalar compares a chunk)
                    if (kc == 0) stage2();
                    else if (kc == 1) stage3();
                    else if (kc == 2) stage4();
                    else stage5();
                }
            }
            if (kc + 1 < nk) {
                chunk(kc & 1, std::true_type(), std::false_type());
                ++kc;
            }
            chunk(kc & 1, std::false_type(), std::false_type());
        };
        if (T.any) {
            prefetch(0);
            sweep(chunk_mma);
        } else sweep(chunk_idle);
        if (!peeled) { stage2(); stage3(); stage4(); stage5(); }

        // ---- epilogue of this tile.  First of all the next tile's first chunk goes on its way into the staging registers (dead since
        // the k loop's last loads) -- in front of this tile's stores in the memory pipeline, not behind them --, and one lane asks for
        // the index of the tile after next.
#ifndef ACX_EFP_GLOAD_UNIT
#define ACX_EFP_GLOAD_UNIT (F16 ? -1 : 3)     /* bf16x3 (nine staging pieces, 36 registers): behind the epilogue's fourth unit, when half of the
                                                 accumulators are dead -- at the head the kernel spills */
#endif
        constexpr int GLOAD_UNIT = ACX_EFP_GLOAD_UNIT;
        auto next_chunk0 = [&]() {
            stage6(t_next);
            if (nok) {
                for9([&](auto p_tag) { gload_piece(p_tag); });
                gload_advance();
            }
        };
        if (GLOAD_UNIT < 0) next_chunk0();
        if (tid == 0) pending = atomicAdd(counter, 1u);
        const float *nrm = s == 0 ? nrm0 : nrm1;
        const float *inv = (CH || s == 0) ? inv0 : inv1;
        const int il = lr, jl = 4 * lk;
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        float nx[NA], sx[NA];
        f32x4 ny[NB], sy[NB];
        int64_t cbase[NA][NB];
        int cpitch[NA][NB], ctn[NA][NB];
        if (T.any) {
            if (!CH) {
#pragma unroll
                for (int a = 0; a < NA; ++a) nx[a] = nrm[T.rowA[a] + il];
#pragma unroll
                for (int b = 0; b < NB; ++b) ny[b] = *reinterpret_cast<const f32x4u *>(nrm + T.rowB[b] + jl);
            }
            if (F16) {
#pragma unroll
                for (int a = 0; a < NA; ++a) sx[a] = inv[T.rowA[a] + il];
#pragma unroll
                for (int b = 0; b < NB; ++b) sy[b] = *reinterpret_cast<const f32x4u *>(inv + T.rowB[b] + jl);
            }
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const EfPair *P = pd + (T.pidx(a, b) < 0 ? 0 : T.pidx(a, b));
                    const int pc = P->pitchC;
                    cbase[a][b] = P->offC + (int64_t)s * P->M * pc + (int64_t)T.localA(a) * pc + T.localB(b);
                    cpitch[a][b] = pc;
                    ctn[a][b] = P->ctN;
                }
        }
        auto value = [&](int a, int b, float (&v)[4]) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float dot = F16 ? (acc[a][b][reg] * sx[a]) * sy[b][reg] : acc[a][b][reg];
                if (CH) v[reg] = 1.0f - dot;
                else {
                    float tq = (nx[a] + ny[b][reg]) - 2.0f * dot;
                    if (tq < 0.0f) tq = 0.0f;
                    v[reg] = ef_sqrt_nonneg(tq);
                }
            }
        };
        auto narrow = [&](int a, int b) {
            if (T.pidx(a, b) < 0) return;
            float v[4];
            value(a, b, v);
            float *cr = scratch + cbase[a][b] + (int64_t)il * cpitch[a][b] + jl;
            if (T.validA(a) == 16 && T.validB(b) == 16) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(cr));
            else if (il < T.validA(a)) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (jl + reg < T.validB(b)) cr[reg] = v[reg];
            }
            if (ctn[a][b] && il < T.validA(a)) {
                const EfPair P = pd[T.pidx(a, b)];
                float *ct = scratch + ef_ct_off(P, s) + (size_t)(T.localB(b) + jl) * P.pitchT + T.localA(a) + il;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (jl + reg < T.validB(b)) ct[(size_t)reg * P.pitchT] = v[reg];
            }
        };
        const int tr = lane >> 3, tc = 4 * (lane & 7);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; b += 2) {
                if (T.any) {
                    const bool wide = T.pidx(a, b) >= 0 && T.pidx(a, b) == T.pidx(a, b + 1) && T.validA(a) == 16 && T.validB(b) == 16 &&
                                      T.validB(b + 1) == 16 && T.localB(b + 1) == T.localB(b) + 16 && !ctn[a][b];
                    if (!wide) { narrow(a, b); narrow(a, b + 1); }
                    else {
                        float v0[4], v1[4];
                        value(a, b, v0);
                        value(a, b + 1, v1);
                        const int wz = (il >> 1) & 7;
                        *reinterpret_cast<float4 *>(Tw + il * EFR_TP + 4 * (lk ^ wz)) = make_float4(v0[0], v0[1], v0[2], v0[3]);
                        *reinterpret_cast<float4 *>(Tw + il * EFR_TP + 4 * ((4 + lk) ^ wz)) = make_float4(v1[0], v1[1], v1[2], v1[3]);
                        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(Tw + tr * EFR_TP + 4 * ((lane & 7) ^ ((tr >> 1) & 7)));
                        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(Tw + (8 + tr) * EFR_TP + 4 * ((lane & 7) ^ (((8 + tr) >> 1) & 7)));
                        float *cr = scratch + cbase[a][b] + (int64_t)tr * cpitch[a][b] + tc;
                        __builtin_nontemporal_store(w0, reinterpret_cast<f32x4 *>(cr));
                        __builtin_nontemporal_store(w1, reinterpret_cast<f32x4 *>(cr + (int64_t)8 * cpitch[a][b]));
                    }
                }
                if (2 * a + b / 2 == GLOAD_UNIT) next_chunk0();
            }
        // hand the index of the tile after next round (wave 0 is done with its turning tile: word 0 is the mailbox)
        if (tid == 0) *mailbox = pending;
        if (!nok) break;                              // (uniform: t_next >= total)
        unpack(t_next, T);
        t_cur = t_next;
    }
}

}  // namespace acx
