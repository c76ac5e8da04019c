// EarlyFusion cross-similarity GEMMs, round 6 experiment (VERDICT r05 item 1 b): the two-term fp16 rectangle kernel with its operands
// staged by LDS-DMA -- global_load_lds_dwordx4: global memory -> LDS without a trip through the register file, no ds_write pass --
// into THREE operand buffers.  ACX_EF_DMA=1 selects it (f16x2 arithmetic, one workgroup per tile); what it measured is in
// profiles/r06_ef.md (b).
//
// Reference: the X.dot(Y.T) of get_csm / get_csm_cosine (acoss/utils/cross_recurrence.py:30-73), as ef_gemm_rect_bf16x3_kernel<CH, 1>
// (ef_kernels.hpp) computes them: same tiles, same LDS image of a chunk (64-byte rows, the four 16-byte pieces of a row XOR-swizzled),
// the same MFMAs in the same order, the same epilogue -- bit-identical matrices.  What differs:
//   * a wave-instruction of the DMA writes 1 KB of LDS LINEARLY (wave-uniform base + 16 x lane): exactly the 16 rows x 64 bytes a
//     wave staged per piece before.  The swizzle therefore moves to the SOURCE: the lane that lands on position `pos` of row r
//     fetches logical piece pos ^ swz(r) (the same involution the operand reads apply);
//   * the data of chunk k + 2 is on its way into buffer (k + 2) % 3 while chunk k is multiplied out of buffer k % 3 and chunk k + 1
//     sits complete in buffer (k + 1) % 3: 3 x 48 KB + the 16 KB of turning tiles = all 160 KB (the three-term bf16 operands would
//     need 216 KB: this kernel exists for the fp16 arithmetic only);
//   * ordering is by hand: a wave waits `vmcnt(6)` -- everything but the six pieces it has just issued -- before the chunk's barrier,
//     and every wave reads the next buffer only behind that barrier (nothing else orders a ds_read behind somebody's DMA); a buffer is
//     refilled a whole barrier after its last read.
#pragma once
#include "ef_kernels.hpp"

namespace acx {

constexpr int EFD_A = 2 * EFR_ROWS * EFB_LP, EFD_B = 2 * EFR_COLS * EFB_LP;      // fp16 elements of one operand buffer (two terms)
static_assert(2 * 3 * (EFD_A + EFD_B) + 8 * 16 * EFR_TP * 4 == EFR_LDS_BYTES, "three buffers and the turning tiles fill the LDS");

template <int CH>
__global__ __launch_bounds__(EFR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ef_gemm_rect_dma_kernel(
    const unsigned short *__restrict__ split0, const unsigned short *__restrict__ split1, const float *__restrict__ nrm0,
    const float *__restrict__ nrm1, const EfPair *__restrict__ pd, const EfSegRect *__restrict__ rects,
    const EfSegWg *__restrict__ wgs, const EfSegGroup *__restrict__ rowg, const EfSegGroup *__restrict__ colg, const int32_t *__restrict__ pairtab,
    float *__restrict__ scratch, int Kp0, int Kp1, const float *__restrict__ inv0, const float *__restrict__ inv1)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short efr_lds[];
    unsigned short *As = efr_lds;                    // [buffer][term][row][32 k]
    unsigned short *Bs = efr_lds + 3 * EFD_A;
    const EfSegWg W = wgs[blockIdx.x];
    const EfSegRect R = rects[W.rect];
    const int ty = W.ty, tx = W.tx, ncg = W.pad;
    const int s = CH ? 2 : (int)blockIdx.z;          // 0 mfcc, 1 ssm, 2 chroma
    const int Kp = (CH || s == 0) ? Kp0 : Kp1;
    const unsigned short *S = (CH || s == 0) ? split0 : split1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int NA = 4, NB = 4, NT = 2;
    const int gr0 = 16 * ty + NA * wr, gc0 = tx + NB * wc;
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
    f32x4 acc[NA][NB];
    float zero_ = 0.0f;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{zero_, zero_, zero_, zero_};

    // staging: lane 4 r + pos of a wave lands on position pos of row r of the wave's 16 rows; it fetches logical piece pos ^ swz(row)
    const int srow = tid >> 2, pos = tid & 3;
    const int sp = pos ^ ((0x78 >> (2 * ((srow >> 2) & 3))) & 3);
    const int sg = wave, sr = srow & 15;
    const int pieces = Kp / 8;
    const unsigned short *ap0 = S, *ap1 = S, *bp = S + sp * 8;
    int sp0 = sp, sp1 = sp;
    {
        int rslot = 0;
        if (CH) rslot = colg[R.h0 + tx].slot;
        auto roll_of = [&](const EfSegGroup &g) {
            const int p = pairtab[R.ptab0 + g.slot * R.ncols + rslot];
            int r = p >= 0 ? pd[p].oti : 0;
            r = (sp - (pieces / 12) * r) % pieces;
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
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void gbl_void_t;
    // piece j of a chunk: 0 / 1 the two terms of A rows tid / 4, 2 / 3 of A rows 128 + tid / 4, 4 / 5 of B rows tid / 4
    auto dma_piece = [&](int buf, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        constexpr int t = j & 1, which = j >> 1;
        const unsigned short *src;
        if (which == 2) src = bp;
        else if (!CH) src = which == 0 ? ap0 : ap1;
        else {
            const int q = which == 0 ? sp0 : sp1;
            src = (which == 0 ? ap0 : ap1) + (32 * NT) * (q >> 2) + 8 * (q & 3);
        }
        src += t * EFB_BK;
        unsigned short *dst = which == 2 ? Bs + buf * EFD_B + (t * EFR_COLS + 16 * wave) * EFB_LP
                                         : As + buf * EFD_A + (t * EFR_ROWS + 128 * which + 16 * wave) * EFB_LP;
        __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)dst, 16, 0, 0);
    };
    auto dma_advance = [&]() {
        bp += NT * EFB_BK;
        if (!CH) { ap0 += NT * EFB_BK; ap1 += NT * EFB_BK; }
        else {
            sp0 += 4; sp0 = sp0 >= pieces ? sp0 - pieces : sp0;
            sp1 += 4; sp1 = sp1 >= pieces ? sp1 - pieces : sp1;
        }
    };
    auto for6 = [&](auto &&f) {
        f(std::integral_constant<int, 0>()); f(std::integral_constant<int, 1>()); f(std::integral_constant<int, 2>());
        f(std::integral_constant<int, 3>()); f(std::integral_constant<int, 4>()); f(std::integral_constant<int, 5>());
    };
    const int lks = lk ^ ((0x78 >> (2 * ((lr >> 2) & 3))) & 3);
    const unsigned short *aop = As + (64 * wr + lr) * EFB_LP + 8 * lks;
    const unsigned short *bop = Bs + (64 * wc + lr) * EFB_LP + 8 * lks;

    bf16x8 pa0[NA], pa1[NA], pb0, pb1;               // prefetched: both terms of the four row sub-tiles and of column sub-tile 0 of the next chunk
    auto prefetch = [&](int buf) {
        const unsigned short *a_ = aop + buf * EFD_A, *b_ = bop + buf * EFD_B;
        pb1 = *reinterpret_cast<const bf16x8 *>(b_ + EFR_COLS * EFB_LP);
#pragma unroll
        for (int a = 0; a < NA; ++a) pa0[a] = *reinterpret_cast<const bf16x8 *>(a_ + (16 * a) * EFB_LP);
        pb0 = *reinterpret_cast<const bf16x8 *>(b_);
#pragma unroll
        for (int a = 0; a < NA; ++a) pa1[a] = *reinterpret_cast<const bf16x8 *>(a_ + (EFR_ROWS + 16 * a) * EFB_LP);
    };
    // one chunk out of buffer `cur`; ST: a next chunk exists (wait for it, barrier, prefetch it); LD: the chunk after next exists (its DMA
    // goes to buffer `nn` behind slots 1 .. 6).  Products in ef_gemm_rect_bf16x3_kernel<CH, 1>'s order: x1 y2, x2 y1, x1 y1.
    auto chunk_mma = [&](int cur, int nxt, int nn, auto st_tag, auto ld_tag) {
        constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
        constexpr int NG = 3;
        constexpr int TA[3] = {0, 1, 0}, TB[3] = {1, 0, 0};
        constexpr int LASTP = 8;
        const unsigned short *b_ = bop + cur * EFD_B;
        bf16x8 av[NA][2], bv[2][2];
        auto rdb = [&](int e, int b, int q) { bv[e][q] = *reinterpret_cast<const bf16x8 *>(b_ + (q * EFR_COLS + 16 * b) * EFB_LP); };
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
                if (LD && slot >= 1 && slot <= 6) {
                    switch (slot - 1) {
                    case 0: dma_piece(nn, std::integral_constant<int, 0>()); break;
                    case 1: dma_piece(nn, std::integral_constant<int, 1>()); break;
                    case 2: dma_piece(nn, std::integral_constant<int, 2>()); break;
                    case 3: dma_piece(nn, std::integral_constant<int, 3>()); break;
                    case 4: dma_piece(nn, std::integral_constant<int, 4>()); break;
                    default: dma_piece(nn, std::integral_constant<int, 5>()); break;
                    }
                }
                if (ST && slot == LASTP) {
                    if (LD) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
                    prefetch(nxt);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (LD) dma_advance();
    };
    auto chunk_idle = [&](int cur, int nxt, int nn, auto st_tag, auto ld_tag) {
        constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
        (void)cur; (void)nxt;
        if (LD) { for6([&](auto j_tag) { dma_piece(nn, j_tag); }); dma_advance(); }
        if (ST) {
            if (LD) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
        }
    };
    const int nk = Kp / EFB_BK;
    // prologue: chunk 0 into buffer 0, chunk 1 into buffer 1; chunk 0 complete and visible before the first read
    for6([&](auto j_tag) { dma_piece(0, j_tag); });
    dma_advance();
    if (nk > 1) {
        for6([&](auto j_tag) { dma_piece(1, j_tag); });
        dma_advance();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
    auto sweep = [&](auto &&chunk) {
        int kc = 0, cur = 0;
        auto inc = [](int b) { return b == 2 ? 0 : b + 1; };
        for (; kc + 2 < nk; ++kc) { chunk(cur, inc(cur), inc(inc(cur)), std::true_type(), std::true_type()); cur = inc(cur); }
        if (kc + 1 < nk) {
            chunk(cur, inc(cur), inc(inc(cur)), std::true_type(), std::false_type());
            cur = inc(cur);
            ++kc;
        }
        chunk(cur, inc(cur), inc(inc(cur)), std::false_type(), std::false_type());
    };
    if (any) {
        prefetch(0);
        sweep(chunk_mma);
    } else sweep(chunk_idle);

    // ---- epilogue: ef_gemm_rect_bf16x3_kernel<CH, 1>'s
    if (!any) return;
    const float *nrm = s == 0 ? nrm0 : nrm1;
    const float *inv = (CH || s == 0) ? inv0 : inv1;
    const int il = lr, jl = 4 * lk;
    float nx[NA];
    f32x4 ny[NB];
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    if (!CH) {
#pragma unroll
        for (int a = 0; a < NA; ++a) nx[a] = nrm[GA[a].poolrow + il];
#pragma unroll
        for (int b = 0; b < NB; ++b) ny[b] = *reinterpret_cast<const f32x4u *>(nrm + GB[b].poolrow + jl);
    }
    float sx[NA];
    f32x4 sy[NB];
#pragma unroll
    for (int a = 0; a < NA; ++a) sx[a] = inv[GA[a].poolrow + il];
#pragma unroll
    for (int b = 0; b < NB; ++b) sy[b] = *reinterpret_cast<const f32x4u *>(inv + GB[b].poolrow + jl);
    int64_t cbase[NA][NB];
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
    float *Tw = reinterpret_cast<float *>(efr_lds + 3 * (EFD_A + EFD_B)) + wave * (16 * EFR_TP);
    auto value = [&](int a, int b, float (&v)[4]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float dot = (acc[a][b][reg] * sx[a]) * sy[b][reg];
            if (CH) v[reg] = 1.0f - dot;
            else {
                float tq = (nx[a] + ny[b][reg]) - 2.0f * dot;
                if (tq < 0.0f) tq = 0.0f;
                v[reg] = ef_sqrt_nonneg(tq);
            }
        }
    };
    auto narrow = [&](int a, int b) {
        if (pidx[a][b] < 0) return;
        float v[4];
        value(a, b, v);
        float *cr = scratch + cbase[a][b] + (int64_t)il * cpitch[a][b] + jl;
        if (GA[a].valid == 16 && GB[b].valid == 16) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(cr));
        else if (il < GA[a].valid) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (jl + reg < GB[b].valid) cr[reg] = v[reg];
        }
        if (ctn[a][b] && il < GA[a].valid) {
            const EfPair P = pd[pidx[a][b]];
            float *ct = scratch + ef_ct_off(P, s) + (size_t)(GB[b].local0 + jl) * P.pitchT + GA[a].local0 + il;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (jl + reg < GB[b].valid) ct[(size_t)reg * P.pitchT] = v[reg];
        }
    };
    const int tr = lane >> 3, tc = 4 * (lane & 7);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; b += 2) {
            const bool wide = pidx[a][b] >= 0 && pidx[a][b] == pidx[a][b + 1] && GA[a].valid == 16 && GB[b].valid == 16 &&
                              GB[b + 1].valid == 16 && GB[b + 1].local0 == GB[b].local0 + 16 && !ctn[a][b];
            if (!wide) { narrow(a, b); narrow(a, b + 1); continue; }
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


// ---- the three-term bf16 arithmetic (all 24 bits of every operand) by LDS-DMA: its two operand buffers are all the LDS there is (2 x 72 KB
// + the turning tiles), so a chunk's nine pieces go into the buffer the PREVIOUS chunk was read from, behind that chunk's barrier -- one
// chunk ahead instead of two; a bf16x3 chunk is 96 MFMAs per wave, twice the fp16 arithmetic's, and gives the DMA the same time to land.
template <int CH>
__global__ __launch_bounds__(EFR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ef_gemm_rect_dma2_kernel(
    const unsigned short *__restrict__ split0, const unsigned short *__restrict__ split1, const float *__restrict__ nrm0,
    const float *__restrict__ nrm1, const EfPair *__restrict__ pd, const EfSegRect *__restrict__ rects,
    const EfSegWg *__restrict__ wgs, const EfSegGroup *__restrict__ rowg, const EfSegGroup *__restrict__ colg, const int32_t *__restrict__ pairtab,
    float *__restrict__ scratch, int Kp0, int Kp1, const float *__restrict__ inv0, const float *__restrict__ inv1)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short efr_lds[];
    unsigned short *As = efr_lds;                    // [buffer][term][row][32 k]
    unsigned short *Bs = efr_lds + 2 * EFR_A;
    const EfSegWg W = wgs[blockIdx.x];
    const EfSegRect R = rects[W.rect];
    const int ty = W.ty, tx = W.tx, ncg = W.pad;
    const int s = CH ? 2 : (int)blockIdx.z;          // 0 mfcc, 1 ssm, 2 chroma
    const int Kp = (CH || s == 0) ? Kp0 : Kp1;
    const unsigned short *S = (CH || s == 0) ? split0 : split1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int NA = 4, NB = 4, NT = 3;
    const int gr0 = 16 * ty + NA * wr, gc0 = tx + NB * wc;
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
    f32x4 acc[NA][NB];
    const float zero_ = 0.0f;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{zero_, zero_, zero_, zero_};

    // staging: lane 4 r + pos of a wave lands on position pos of row r of the wave's 16 rows; it fetches logical piece pos ^ swz(row)
    const int srow = tid >> 2, pos = tid & 3;
    const int sp = pos ^ ((0x78 >> (2 * ((srow >> 2) & 3))) & 3);
    const int sg = wave, sr = srow & 15;
    const int pieces = Kp / 8;
    const unsigned short *ap0 = S, *ap1 = S, *bp = S + sp * 8;
    int sp0 = sp, sp1 = sp;
    {
        int rslot = 0;
        if (CH) rslot = colg[R.h0 + tx].slot;
        auto roll_of = [&](const EfSegGroup &g) {
            const int p = pairtab[R.ptab0 + g.slot * R.ncols + rslot];
            int r = p >= 0 ? pd[p].oti : 0;
            r = (sp - (pieces / 12) * r) % pieces;
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
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void gbl_void_t;
    // piece j of a chunk: 0-2 the three terms of A rows tid / 4, 3-5 of A rows 128 + tid / 4, 6-8 of B rows tid / 4
    auto dma_piece = [&](int buf, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        constexpr int t = j % 3, which = j / 3;
        const unsigned short *src;
        if (which == 2) src = bp;
        else if (!CH) src = which == 0 ? ap0 : ap1;
        else {
            const int q = which == 0 ? sp0 : sp1;
            src = (which == 0 ? ap0 : ap1) + (32 * NT) * (q >> 2) + 8 * (q & 3);
        }
        src += t * EFB_BK;
        unsigned short *dst = which == 2 ? Bs + buf * EFR_B + (t * EFR_COLS + 16 * wave) * EFB_LP
                                         : As + buf * EFR_A + (t * EFR_ROWS + 128 * which + 16 * wave) * EFB_LP;
        __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)dst, 16, 0, 0);
    };
    auto dma_advance = [&]() {
        bp += NT * EFB_BK;
        if (!CH) { ap0 += NT * EFB_BK; ap1 += NT * EFB_BK; }
        else {
            sp0 += 4; sp0 = sp0 >= pieces ? sp0 - pieces : sp0;
            sp1 += 4; sp1 = sp1 >= pieces ? sp1 - pieces : sp1;
        }
    };
    auto for9 = [&](auto &&f) {
        f(std::integral_constant<int, 0>()); f(std::integral_constant<int, 1>()); f(std::integral_constant<int, 2>());
        f(std::integral_constant<int, 3>()); f(std::integral_constant<int, 4>()); f(std::integral_constant<int, 5>());
        f(std::integral_constant<int, 6>()); f(std::integral_constant<int, 7>()); f(std::integral_constant<int, 8>());
    };
    const int lks = lk ^ ((0x78 >> (2 * ((lr >> 2) & 3))) & 3);
    const unsigned short *aop = As + (64 * wr + lr) * EFB_LP + 8 * lks;
    const unsigned short *bop = Bs + (64 * wc + lr) * EFB_LP + 8 * lks;

    bf16x8 pa0[NA], pa2[NA], pb0, pb2;               // prefetched: terms 0 and 2 of the four row sub-tiles and of column sub-tile 0 of the next chunk
    auto prefetch = [&](int buf) {
        const unsigned short *a_ = aop + buf * EFR_A, *b_ = bop + buf * EFR_B;
        pb2 = *reinterpret_cast<const bf16x8 *>(b_ + (2 * EFR_COLS) * EFB_LP);
#pragma unroll
        for (int a = 0; a < NA; ++a) pa0[a] = *reinterpret_cast<const bf16x8 *>(a_ + (16 * a) * EFB_LP);
        pb0 = *reinterpret_cast<const bf16x8 *>(b_);
#pragma unroll
        for (int a = 0; a < NA; ++a) pa2[a] = *reinterpret_cast<const bf16x8 *>(a_ + (2 * EFR_ROWS + 16 * a) * EFB_LP);
    };
    // one chunk out of buffer `cur`; ST: a next chunk exists -- its nine pieces go by DMA into the OTHER buffer behind slots 1 .. 9 (every
    // wave is past the previous chunk's barrier, i.e. past its last read of that buffer), the wave waits for them (vmcnt(0): nothing
    // younger is in flight) before this chunk's barrier behind slot 18, and reads them behind it.  Products in
    // ef_gemm_rect_bf16x3_kernel<CH, 0>'s order: x1 y3, x3 y1, x2 y2, x1 y2, x2 y1, x1 y1.
    auto chunk_mma = [&](int cur, auto st_tag) {
        constexpr bool ST = decltype(st_tag)::value;
        constexpr int NG = 6;
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
        constexpr int LASTP = 18;
        const unsigned short *a_ = aop + cur * EFR_A, *b_ = bop + cur * EFR_B;
        bf16x8 av[NA][3], bv[2][3];
        auto rdb = [&](int e, int b, int q) { bv[e][q] = *reinterpret_cast<const bf16x8 *>(b_ + (q * EFR_COLS + 16 * b) * EFB_LP); };
#pragma unroll
        for (int a = 0; a < NA; ++a) { av[a][0] = pa0[a]; av[a][2] = pa2[a]; }
        bv[0][2] = pb2; bv[0][0] = pb0;
        rdb(0, 0, 1);
#pragma unroll
        for (int a = 0; a < NA; ++a) av[a][1] = *reinterpret_cast<const bf16x8 *>(a_ + (EFR_ROWS + 16 * a) * EFB_LP);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv[b & 1][TB[g]], av[a][TA[g]], acc[a][b], 0, 0, 0);
                if (g == 1 && b + 1 < NB) { rdb((b + 1) & 1, b + 1, 2); rdb((b + 1) & 1, b + 1, 0); rdb((b + 1) & 1, b + 1, 1); }
                const int slot = NG * b + g;
                if (ST && slot >= 1 && slot <= 9) {
                    switch (slot - 1) {
                    case 0: dma_piece(cur ^ 1, std::integral_constant<int, 0>()); break;
                    case 1: dma_piece(cur ^ 1, std::integral_constant<int, 1>()); break;
                    case 2: dma_piece(cur ^ 1, std::integral_constant<int, 2>()); break;
                    case 3: dma_piece(cur ^ 1, std::integral_constant<int, 3>()); break;
                    case 4: dma_piece(cur ^ 1, std::integral_constant<int, 4>()); break;
                    case 5: dma_piece(cur ^ 1, std::integral_constant<int, 5>()); break;
                    case 6: dma_piece(cur ^ 1, std::integral_constant<int, 6>()); break;
                    case 7: dma_piece(cur ^ 1, std::integral_constant<int, 7>()); break;
                    default: dma_piece(cur ^ 1, std::integral_constant<int, 8>()); break;
                    }
                }
                if (ST && slot == LASTP) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
                    prefetch(cur ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ST) dma_advance();
    };
    auto chunk_idle = [&](int cur, auto st_tag) {
        constexpr bool ST = decltype(st_tag)::value;
        if (ST) {
            for9([&](auto j_tag) { dma_piece(cur ^ 1, j_tag); });
            dma_advance();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
        }
    };
    const int nk = Kp / EFB_BK;
    // prologue: chunk 0 into buffer 0, complete and visible before the first read (chunk 1 follows behind chunk 0's first MFMAs)
    for9([&](auto j_tag) { dma_piece(0, j_tag); });
    dma_advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
    auto sweep = [&](auto &&chunk) {
        int kc = 0;
        for (; kc + 1 < nk; ++kc) chunk(kc & 1, std::true_type());
        chunk(kc & 1, std::false_type());
    };
    if (any) {
        prefetch(0);
        sweep(chunk_mma);
    } else sweep(chunk_idle);

    // ---- epilogue: ef_gemm_rect_bf16x3_kernel<CH, 0>'s
    if (!any) return;
    const float *nrm = s == 0 ? nrm0 : nrm1;
    const int il = lr, jl = 4 * lk;
    float nx[NA];
    f32x4 ny[NB];
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    if (!CH) {
#pragma unroll
        for (int a = 0; a < NA; ++a) nx[a] = nrm[GA[a].poolrow + il];
#pragma unroll
        for (int b = 0; b < NB; ++b) ny[b] = *reinterpret_cast<const f32x4u *>(nrm + GB[b].poolrow + jl);
    }
    int64_t cbase[NA][NB];
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
    float *Tw = reinterpret_cast<float *>(efr_lds + 2 * (EFR_A + EFR_B)) + wave * (16 * EFR_TP);
    auto value = [&](int a, int b, float (&v)[4]) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float dot = acc[a][b][reg];
            if (CH) v[reg] = 1.0f - dot;
            else {
                float tq = (nx[a] + ny[b][reg]) - 2.0f * dot;
                if (tq < 0.0f) tq = 0.0f;
                v[reg] = ef_sqrt_nonneg(tq);
            }
        }
    };
    auto narrow = [&](int a, int b) {
        if (pidx[a][b] < 0) return;
        float v[4];
        value(a, b, v);
        float *cr = scratch + cbase[a][b] + (int64_t)il * cpitch[a][b] + jl;
        if (GA[a].valid == 16 && GB[b].valid == 16) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(cr));
        else if (il < GA[a].valid) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (jl + reg < GB[b].valid) cr[reg] = v[reg];
        }
        if (ctn[a][b] && il < GA[a].valid) {
            const EfPair P = pd[pidx[a][b]];
            float *ct = scratch + ef_ct_off(P, s) + (size_t)(GB[b].local0 + jl) * P.pitchT + GA[a].local0 + il;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                if (jl + reg < GB[b].valid) ct[(size_t)reg * P.pitchT] = v[reg];
        }
    };
    const int tr = lane >> 3, tc = 4 * (lane & 7);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; b += 2) {
            const bool wide = pidx[a][b] >= 0 && pidx[a][b] == pidx[a][b + 1] && GA[a].valid == 16 && GB[b].valid == 16 &&
                              GB[b + 1].valid == 16 && GB[b + 1].local0 == GB[b].local0 + 16 && !ctn[a][b];
            if (!wide) { narrow(a, b); narrow(a, b + 1); continue; }
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



// ---- the same staging inside the PERSISTENT workgroup of ef_gemm_persist_kernels.hpp (fp16 arithmetic): what round 6 ships as the default.
// With the operands going straight to LDS a tile boundary shrinks to ONE barrier: at the head of a tile's epilogue the next tile's chunks 0
// and 1 go by DMA into the two buffers the k loop's last two barriers have freed (the buffer after the last chunk's, and the one after
// that), in FRONT of the epilogue's own loads -- whose arrival, in order, tells the wave that its pieces have landed -- and of its stores;
// behind the boundary's barrier (which also hands the next tile index round) the k loop starts at once.
template <int CH>
__global__ __launch_bounds__(EFR_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ef_gemm_rect_persist_dma_kernel(
    const unsigned short *__restrict__ split0, const unsigned short *__restrict__ split1, const float *__restrict__ nrm0,
    const float *__restrict__ nrm1, const EfPair *__restrict__ pd, const EfSegRect *__restrict__ rects,
    const EfSegWg *__restrict__ wgs, const EfSegGroup *__restrict__ rowg, const EfSegGroup *__restrict__ colg, const int32_t *__restrict__ pairtab,
    float *__restrict__ scratch, int Kp0, int Kp1, const float *__restrict__ inv0, const float *__restrict__ inv1,
    int ntiles, int nfeat, unsigned *__restrict__ counter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short efr_lds[];
    unsigned short *As = efr_lds;                    // [buffer][term][row][32 k]
    unsigned short *Bs = efr_lds + 3 * EFD_A;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 15, lk = lane >> 4;
    constexpr int NA = 4, NB = 4, NT = 2;
    constexpr int F16 = 1;
    const int total = ntiles * nfeat;
    int t_cur = (int)blockIdx.x, t_next = (int)(blockIdx.x + gridDim.x);
    if (t_cur >= total) return;
    float *Tw = reinterpret_cast<float *>(efr_lds + 3 * (EFD_A + EFD_B)) + wave * (16 * EFR_TP);
    volatile unsigned *mailbox = reinterpret_cast<volatile unsigned *>(efr_lds + 3 * (EFD_A + EFD_B));      // (word 0 of wave 0's turning tile)
    // lane 4 r + pos of a wave lands on position pos of row r of the wave's 16 rows: it fetches logical piece sp = pos ^ swz(row)
    const int srow = tid >> 2;
    const int sp = (tid & 3) ^ ((0x78 >> (2 * ((srow >> 2) & 3))) & 3);
    const int sg = wave, sr = srow & 15;
    f32x4 acc[NA][NB];

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

    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef __attribute__((address_space(1))) const void gbl_void_t;
    // piece j of a chunk: 0 / 1 the two terms of A rows tid / 4, 2 / 3 of A rows 128 + tid / 4, 4 / 5 of B rows tid / 4
    auto dma_piece = [&](int buf, auto j_tag) {
        constexpr int j = decltype(j_tag)::value;
        constexpr int t = j & 1, which = j >> 1;
        const unsigned short *src;
        if (which == 2) src = bp;
        else if (!CH) src = which == 0 ? ap0 : ap1;
        else {
            const int q = which == 0 ? sp0 : sp1;
            src = (which == 0 ? ap0 : ap1) + (32 * NT) * (q >> 2) + 8 * (q & 3);
        }
        src += t * EFB_BK;
        unsigned short *dst = which == 2 ? Bs + buf * EFD_B + (t * EFR_COLS + 16 * wave) * EFB_LP
                                         : As + buf * EFD_A + (t * EFR_ROWS + 128 * which + 16 * wave) * EFB_LP;
        __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)dst, 16, 0, 0);
    };
    auto dma_advance = [&]() {
        bp += NT * EFB_BK;
        if (!CH) { ap0 += NT * EFB_BK; ap1 += NT * EFB_BK; }
        else {
            sp0 += 4; sp0 = sp0 >= pieces ? sp0 - pieces : sp0;
            sp1 += 4; sp1 = sp1 >= pieces ? sp1 - pieces : sp1;
        }
    };
    auto for6 = [&](auto &&f) {
        f(std::integral_constant<int, 0>()); f(std::integral_constant<int, 1>()); f(std::integral_constant<int, 2>());
        f(std::integral_constant<int, 3>()); f(std::integral_constant<int, 4>()); f(std::integral_constant<int, 5>());
    };
    const int lks = lk ^ ((0x78 >> (2 * ((lr >> 2) & 3))) & 3);
    const unsigned short *aop = As + (64 * wr + lr) * EFB_LP + 8 * lks;
    const unsigned short *bop = Bs + (64 * wc + lr) * EFB_LP + 8 * lks;

    bf16x8 pa0[NA], pa1[NA], pb0, pb1;               // prefetched: both terms of the four row sub-tiles and of column sub-tile 0 of the next chunk
    auto prefetch = [&](int buf) {
        const unsigned short *a_ = aop + buf * EFD_A, *b_ = bop + buf * EFD_B;
        pb1 = *reinterpret_cast<const bf16x8 *>(b_ + EFR_COLS * EFB_LP);
#pragma unroll
        for (int a = 0; a < NA; ++a) pa0[a] = *reinterpret_cast<const bf16x8 *>(a_ + (16 * a) * EFB_LP);
        pb0 = *reinterpret_cast<const bf16x8 *>(b_);
#pragma unroll
        for (int a = 0; a < NA; ++a) pa1[a] = *reinterpret_cast<const bf16x8 *>(a_ + (EFR_ROWS + 16 * a) * EFB_LP);
    };
    // one chunk out of buffer `cur`; ST: a next chunk exists (wait for it, barrier, prefetch it); LD: the chunk after next exists (its DMA
    // goes to buffer `nn` behind slots 1 .. 6).  Products in ef_gemm_rect_bf16x3_kernel<CH, 1>'s order: x1 y2, x2 y1, x1 y1.
    auto chunk_mma = [&](int cur, int nxt, int nn, auto st_tag, auto ld_tag) {
        constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
        constexpr int NG = 3;
        constexpr int TA[3] = {0, 1, 0}, TB[3] = {1, 0, 0};
        constexpr int LASTP = 8;
        const unsigned short *b_ = bop + cur * EFD_B;
        bf16x8 av[NA][2], bv[2][2];
        auto rdb = [&](int e, int b, int q) { bv[e][q] = *reinterpret_cast<const bf16x8 *>(b_ + (q * EFR_COLS + 16 * b) * EFB_LP); };
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
                if (LD && slot >= 1 && slot <= 6) {
                    switch (slot - 1) {
                    case 0: dma_piece(nn, std::integral_constant<int, 0>()); break;
                    case 1: dma_piece(nn, std::integral_constant<int, 1>()); break;
                    case 2: dma_piece(nn, std::integral_constant<int, 2>()); break;
                    case 3: dma_piece(nn, std::integral_constant<int, 3>()); break;
                    case 4: dma_piece(nn, std::integral_constant<int, 4>()); break;
                    default: dma_piece(nn, std::integral_constant<int, 5>()); break;
                    }
                }
                if (ST && slot == LASTP) {
                    if (LD) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
                    prefetch(nxt);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (LD) dma_advance();
    };
    auto chunk_idle = [&](int cur, int nxt, int nn, auto st_tag, auto ld_tag) {
        constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
        (void)cur; (void)nxt;
        if (LD) { for6([&](auto j_tag) { dma_piece(nn, j_tag); }); dma_advance(); }
        if (ST) {
            if (LD) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
        }
    };

    // ---- the first tile: resolved the way the one-tile kernel starts; its chunk 0 into buffer 0
    Tile T;
    stage1(t_cur); stage2(); stage3(); stage4(); stage5();
    unpack(t_cur, T);
    stage6(t_cur);
    int base = 0;                                     // the buffer of the current tile's chunk 0 (chunk j: buffer (base + j) % 3)
    auto inc = [](int b) { return b == 2 ? 0 : b + 1; };
    for6([&](auto j_tag) { dma_piece(0, j_tag); });
    dma_advance();
    unsigned pending = 0;                             // (lane 0 of wave 0: the index fetched for the tile after next)

    for (bool first = true;; first = false) {
        const int s = T.s;
        const int Kp = kp_of(s);
        const int nk = Kp / EFB_BK;
        if (first) {                                  // (later tiles: both chunks were sent at the head of the previous epilogue)
            if (nk > 1) {
                for6([&](auto j_tag) { dma_piece(1, j_tag); });
                dma_advance();
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // every wave's pieces of chunk 0 have landed (its own: see the epilogue), every wave is out of the previous tile's k loop
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");          // (the builtin is not a compiler-level memory barrier: nothing below may be hoisted above it)
        if (!first) t_next = (int)__builtin_amdgcn_readfirstlane(*mailbox);
        float zero_ = 0.0f;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{zero_, zero_, zero_, zero_};
        stage1(t_next);
        const bool peeled = nk >= 7;
        int cur = base;
        auto sweep = [&](auto &&chunk) {
            int kc = 0;
            for (; kc + 2 < nk; ++kc) {
                chunk(cur, inc(cur), inc(inc(cur)), std::true_type(), std::true_type());
                cur = inc(cur);
                if (peeled && kc < 4) {
                    if (kc == 0) stage2();
                    else if (kc == 1) stage3();
                    else if (kc == 2) stage4();
                    else stage5();
                }
            }
            if (kc + 1 < nk) {
                chunk(cur, inc(cur), inc(inc(cur)), std::true_type(), std::false_type());
                cur = inc(cur);
                ++kc;
            }
            chunk(cur, inc(cur), inc(inc(cur)), std::false_type(), std::false_type());
        };
        if (T.any) {
            prefetch(base);
            sweep(chunk_mma);
        } else sweep(chunk_idle);
        if (!peeled) { stage2(); stage3(); stage4(); stage5(); }

        // ---- head of the epilogue: the next tile's chunks 0 and 1 into the two free buffers (`cur` is the last chunk's)
        const int nbase = inc(cur);
        const int nk_next = kp_of(feat_of(t_next)) / EFB_BK;
        stage6(t_next);
        if (nok) {
            for6([&](auto j_tag) { dma_piece(nbase, j_tag); });
            dma_advance();
            if (nk_next > 1) {
                for6([&](auto j_tag) { dma_piece(inc(nbase), j_tag); });
                dma_advance();
            }
        }
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
            }
        // a wave without a pair has loaded nothing whose arrival would tell it that its pieces have landed
        if (!T.any && nok) {
            if (nk_next > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // hand the index of the tile after next round (wave 0 is done with its turning tile: word 0 is the mailbox)
        if (tid == 0) *mailbox = pending;
        if (!nok) break;                              // (uniform: t_next >= total)
        unpack(t_next, T);
        t_cur = t_next;
        base = nbase;
    }
}

}  // namespace acx
