// How many wait states does gfx950 need between a 16-bit-input K = 32 MFMA and an instruction that consumes its result?
// (Round 4 saw v_mfma_f32_16x16x16_f16 chained behind v_mfma_f32_16x16x32_f16 return garbage in compiled kernels and a
// destination "over the second source operand" return that operand's bits: scripts/ubench/mfma_overlap_probe.hip shows the
// overlap itself is harmless -- stale destination registers read too early would look exactly like that.)
// Every case is inline asm on FIXED registers with an explicit number of wait states W between producer and consumer (W = 0:
// back to back); the reference is the same sequence with 34 wait states.  Prints, per consumer kind, the smallest W from which on
// every larger W is correct -- what the compiler's hazard recogniser must insert (scripts/isa_lint.py checks that it did).
//   hipcc --offload-arch=gfx950 -O2 -w -o build_ab/mfma_waitstate_probe scripts/ubench/mfma_waitstate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// producer: P_ v[56:59] = A v[40:43] x B v[44:47] + 0.   second operands: A2 v[60:63], B2 v[64:67].   W_: the wait-state text.
#define SETUP                                                                                                              \
    "v_mov_b32 v40, %[a0]\n\tv_mov_b32 v41, %[a1]\n\tv_mov_b32 v42, %[a2]\n\tv_mov_b32 v43, %[a3]\n\t"                       \
    "v_mov_b32 v44, %[b0]\n\tv_mov_b32 v45, %[b1]\n\tv_mov_b32 v46, %[b2]\n\tv_mov_b32 v47, %[b3]\n\t"                       \
    "v_mov_b32 v60, %[b0]\n\tv_mov_b32 v61, %[a1]\n\tv_mov_b32 v62, %[b2]\n\tv_mov_b32 v63, %[a3]\n\t"                       \
    "v_mov_b32 v64, %[a0]\n\tv_mov_b32 v65, %[b1]\n\tv_mov_b32 v66, %[a2]\n\tv_mov_b32 v67, %[b3]\n\t"                       \
    "v_mov_b32 v56, 0x7fc00000\n\tv_mov_b32 v57, 0x7fc00000\n\tv_mov_b32 v58, 0x7fc00000\n\tv_mov_b32 v59, 0x7fc00000\n\t"   \
    "v_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\t"                                       \
    "s_nop 15\n\t"
#define DRAIN "s_nop 15\n\ts_nop 15\n\ts_nop 1\n\t"
#define RUN(P_, W_, CONS_, R0_, R1_, R2_, R3_)                                                                             \
    asm volatile(SETUP P_ " v[56:59], v[40:43], v[44:47], 0\n\t" W_ CONS_ "\n\t" DRAIN                                       \
                 "v_mov_b32 %[o0], " R0_ "\n\tv_mov_b32 %[o1], " R1_ "\n\tv_mov_b32 %[o2], " R2_ "\n\tv_mov_b32 %[o3], " R3_ "\n\t" \
                 : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3])                                    \
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]) \
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",  \
                   "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");                                                   \
    out[(size_t)(slot++) * 64 + l] = o;

// consumers
#define C_SAME(P_) P_ " v[56:59], v[60:63], v[64:67], v[56:59]"                           /* same opcode, accumulate in place */
#define C_K16F16 "v_mfma_f32_16x16x16_f16 v[56:59], v[60:61], v[64:65], v[56:59]"          /* other 16-bit opcode, srcC */
#define C_F32 "v_mfma_f32_16x16x4_f32 v[56:59], v60, v64, v[56:59]"                        /* f32 MFMA, srcC */
#define C_SRCA(P_) P_ " v[68:71], v[56:59], v[64:67], 0"                                  /* result as the next MFMA's srcA */
#define C_VALU "v_add_f32 v68, v56, v56\n\tv_add_f32 v69, v57, v57\n\tv_add_f32 v70, v58, v58\n\tv_add_f32 v71, v59, v59"   /* VALU read */
#define C_WAW "v_mov_b32 v56, 1.0\n\tv_mov_b32 v57, 1.0\n\tv_mov_b32 v58, 1.0\n\tv_mov_b32 v59, 1.0"                           /* VALU overwrites the destination */
#define C_WAWM_SAME(P_) P_ " v[56:59], v[60:63], v[64:67], 0"                             /* same opcode overwrites the destination (no read) */
#define C_WAWM_K16 "v_mfma_f32_16x16x16_f16 v[56:59], v[60:61], v[64:65], 0"               /* other 16-bit opcode overwrites the destination */
#define C_WAWM_K16_PART "v_mfma_f32_16x16x16_f16 v[54:57], v[60:61], v[64:65], 0"          /* ... half of it */
#define C_WAWM_F32 "v_mfma_f32_16x16x4_f32 v[56:59], v60, v64, 0"                          /* f32 MFMA overwrites the destination */
#define C_WAR "v_mov_b32 v44, 1.0\n\tv_mov_b32 v45, 1.0\n\tv_mov_b32 v46, 1.0\n\tv_mov_b32 v47, 1.0"                           /* VALU overwrites srcB */

#define W0 ""
#define W1 "s_nop 0\n\t"
#define W2 "s_nop 1\n\t"
#define W3 "s_nop 2\n\t"
#define W4 "s_nop 3\n\t"
#define W5 "s_nop 4\n\t"
#define W6 "s_nop 5\n\t"
#define W7 "s_nop 6\n\t"
#define W8 "s_nop 7\n\t"
#define W9 "s_nop 8\n\t"
#define W10 "s_nop 9\n\t"
#define W11 "s_nop 10\n\t"
#define W12 "s_nop 11\n\t"
#define W14 "s_nop 13\n\t"
#define W16 "s_nop 15\n\t"
#define W18 "s_nop 15\n\ts_nop 1\n\t"
#define W20 "s_nop 15\n\ts_nop 3\n\t"
#define W34 "s_nop 15\n\ts_nop 15\n\ts_nop 1\n\t"
#define ALLW(P_, CONS_, R0_, R1_, R2_, R3_)                                                                                 \
    RUN(P_, W34, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W0, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W1, CONS_, R0_, R1_, R2_, R3_)     \
    RUN(P_, W2, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W3, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W4, CONS_, R0_, R1_, R2_, R3_)      \
    RUN(P_, W5, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W6, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W7, CONS_, R0_, R1_, R2_, R3_)      \
    RUN(P_, W8, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W9, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W10, CONS_, R0_, R1_, R2_, R3_)     \
    RUN(P_, W11, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W12, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W14, CONS_, R0_, R1_, R2_, R3_)   \
    RUN(P_, W16, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W18, CONS_, R0_, R1_, R2_, R3_) RUN(P_, W20, CONS_, R0_, R1_, R2_, R3_)
static const int WS[18] = {34, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 18, 20};
#define KINDS(P_)                                                                                                           \
    ALLW(P_, C_SAME(P_), "v56", "v57", "v58", "v59") ALLW(P_, C_K16F16, "v56", "v57", "v58", "v59")                          \
    ALLW(P_, C_F32, "v56", "v57", "v58", "v59") ALLW(P_, C_SRCA(P_), "v68", "v69", "v70", "v71")                             \
    ALLW(P_, C_VALU, "v68", "v69", "v70", "v71") ALLW(P_, C_WAW, "v56", "v57", "v58", "v59") ALLW(P_, C_WAR, "v56", "v57", "v58", "v59")  \
    ALLW(P_, C_WAWM_SAME(P_), "v56", "v57", "v58", "v59") ALLW(P_, C_WAWM_K16, "v56", "v57", "v58", "v59")                          \
    ALLW(P_, C_WAWM_K16_PART, "v56", "v57", "v58", "v59") ALLW(P_, C_WAWM_F32, "v56", "v57", "v58", "v59")
static const char *KIND[11] = {"same opcode, srcC = result (accumulate in place)", "v_mfma_f32_16x16x16_f16, srcC = result", "v_mfma_f32_16x16x4_f32, srcC = result",
                              "same opcode, srcA = result", "VALU reads the result", "VALU overwrites the destination (WAW)", "VALU overwrites srcB (WAR)",
                              "same opcode overwrites the destination", "v_mfma_f32_16x16x16_f16 overwrites the destination", "v_mfma_f32_16x16x16_f16 overwrites half of it",
                              "v_mfma_f32_16x16x4_f32 overwrites the destination"};
constexpr int NK = 11;

__global__ void probe(const u32x4 *A, const u32x4 *B, f32x4 *out)
{
    const int l = threadIdx.x;
    const u32x4 a = A[blockIdx.x * 64 + l], b = B[blockIdx.x * 64 + l];
    f32x4 o;
    int slot = blockIdx.x * 11 * 18;
    if (blockIdx.x == 0) { KINDS("v_mfma_f32_16x16x32_f16") }
    else { KINDS("v_mfma_f32_16x16x32_bf16") }
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

int main()
{
    std::vector<unsigned> A(2 * 64 * 4), B(2 * 64 * 4);
    unsigned s = 11u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 24) - 128) / 32.0f; };
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64 * 4; ++i) {
            const float a0 = rnd(), a1 = rnd(), b0 = rnd(), b1 = rnd();
            A[t * 256 + i] = t == 0 ? (f2h(a0) | ((unsigned)f2h(a1) << 16)) : (f2bf(a0) | ((unsigned)f2bf(a1) << 16));
            B[t * 256 + i] = t == 0 ? (f2h(b0) | ((unsigned)f2h(b1) << 16)) : (f2bf(b0) | ((unsigned)f2bf(b1) << 16));
        }
    unsigned *dA, *dB; float *dO;
    const size_t nslots = 2 * 11 * 18;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, nslots * 64 * 16);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dO, 0, nslots * 64 * 16);
    hipLaunchKernelGGL(probe, dim3(2), dim3(64), 0, 0, (const u32x4 *)dA, (const u32x4 *)dB, (f32x4 *)dO);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> O(nslots * 64 * 4);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < 2; ++t)
        for (int k = 0; k < NK; ++k) {
            const float *ref = &O[(size_t)((t * NK + k) * 18) * 256];
            int need = 0;
            printf("%-26s -> %-50s wrong at W =", t ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x32_f16", KIND[k]);
            for (int w = 1; w < 18; ++w) {
                const float *g = &O[(size_t)((t * NK + k) * 18 + w) * 256];
                int bad = 0;
                for (int i = 0; i < 256; ++i) bad += memcmp(&g[i], &ref[i], 4) != 0;
                if (bad) { printf(" %d", WS[w]); need = w + 1 < 18 ? WS[w + 1] : 99; }
            }
            printf("%s   => needs >= %d wait states\n", need ? "" : " (never)", need);
        }
    return 0;
}
