// Which register overlaps does a 16-bit-input K = 32 MFMA survive on gfx950?  (The rule scripts/isa_lint.py encodes.)
// Every case places the operands in FIXED physical registers through inline asm -- nothing is left to the register allocator --
// and compares the 16 x 16 result with the same instruction on disjoint registers.
//   cases: opcode {f16, bf16} x destination {disjoint, == srcA, == srcB, upper half over srcA, upper half over srcB} x accumulator {constant 0, a register
//   of its own, the destination}.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_overlap_probe scripts/ubench/mfma_overlap_probe.hip && /tmp/mfma_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A in v[40:43], B in v[44:47], C in v[48:51]; D_ = destination, CS_ = the srcC operand text
#define CASE(OP_, D_, D0_, D1_, D2_, D3_, CS_)                                                                                  \
    asm volatile("v_mov_b32 v40, %[a0]\n\tv_mov_b32 v41, %[a1]\n\tv_mov_b32 v42, %[a2]\n\tv_mov_b32 v43, %[a3]\n\t"               \
                 "v_mov_b32 v44, %[b0]\n\tv_mov_b32 v45, %[b1]\n\tv_mov_b32 v46, %[b2]\n\tv_mov_b32 v47, %[b3]\n\t"               \
                 "v_mov_b32 v48, %[c0]\n\tv_mov_b32 v49, %[c1]\n\tv_mov_b32 v50, %[c2]\n\tv_mov_b32 v51, %[c3]\n\t"               \
                 "v_mov_b32 v52, %[c0]\n\tv_mov_b32 v53, %[c1]\n\tv_mov_b32 v54, %[c2]\n\tv_mov_b32 v55, %[c3]\n\t"               \
                 "s_nop 7\n\t" OP_ " " D_ ", v[40:43], v[44:47], " CS_ "\n\t"                                                     \
                 "s_nop 15\n\ts_nop 15\n\t"                                                                                       \
                 "v_mov_b32 %[o0], " D0_ "\n\tv_mov_b32 %[o1], " D1_ "\n\tv_mov_b32 %[o2], " D2_ "\n\tv_mov_b32 %[o3], " D3_ "\n\t" \
                 : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3])                                         \
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),  \
                   [b3] "v"(b[3]), [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3])                                 \
                 : "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
                   "v56", "v57", "v58", "v59")

// accumulator forms: "0" (constant), "v[48:51]" (a register of its own), or the destination (only where the destination is
// not a source: in-place accumulation over a source would need C's values there, which is a different experiment)
#define DESTS(OP_, CS_, base_)                                                                                                   \
    CASE(OP_, "v[56:59]", "v56", "v57", "v58", "v59", CS_); store(base_ + 0);                                                     \
    CASE(OP_, "v[40:43]", "v40", "v41", "v42", "v43", CS_); store(base_ + 1);                                                     \
    CASE(OP_, "v[44:47]", "v44", "v45", "v46", "v47", CS_); store(base_ + 2);                                                     \
    CASE(OP_, "v[38:41]", "v38", "v39", "v40", "v41", CS_); store(base_ + 3);                                                     \
    CASE(OP_, "v[42:45]", "v42", "v43", "v44", "v45", CS_); store(base_ + 4);

__global__ void probe(const u32x4 *A, const u32x4 *B, const f32x4 *C, f32x4 *out)
{
    const int l = threadIdx.x;
    const u32x4 a = A[blockIdx.x * 64 + l], b = B[blockIdx.x * 64 + l];
    const f32x4 c = C[l];
    f32x4 o;
    int slot = 0;
    auto store = [&](int k) { out[(blockIdx.x * 32 + k) * 64 + l] = o; (void)slot; };
    if (blockIdx.x == 0) {
        DESTS("v_mfma_f32_16x16x32_f16", "0", 0)
        DESTS("v_mfma_f32_16x16x32_f16", "v[48:51]", 5)
        CASE("v_mfma_f32_16x16x32_f16", "v[52:55]", "v52", "v53", "v54", "v55", "v[52:55]"); store(10);
    } else {
        DESTS("v_mfma_f32_16x16x32_bf16", "0", 0)
        DESTS("v_mfma_f32_16x16x32_bf16", "v[48:51]", 5)
        CASE("v_mfma_f32_16x16x32_bf16", "v[52:55]", "v52", "v53", "v54", "v55", "v[52:55]"); store(10);
    }
}

static unsigned short f2h(float f)       // fp16 of a small exactly representable value
{
    _Float16 h = (_Float16)f;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

int main()
{
    std::vector<unsigned> A(2 * 64 * 4), B(2 * 64 * 4);
    std::vector<float> C(64 * 4);
    unsigned s = 11u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 24) - 128) / 32.0f; };     // multiples of 1/32 in [-4, 4): exact in fp16 and bf16
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64 * 4; ++i) {
            const float a0 = rnd(), a1 = rnd(), b0 = rnd(), b1 = rnd();
            A[t * 256 + i] = t == 0 ? (f2h(a0) | ((unsigned)f2h(a1) << 16)) : (f2bf(a0) | ((unsigned)f2bf(a1) << 16));
            B[t * 256 + i] = t == 0 ? (f2h(b0) | ((unsigned)f2h(b1) << 16)) : (f2bf(b0) | ((unsigned)f2bf(b1) << 16));
        }
    for (int i = 0; i < 256; ++i) C[i] = rnd() * 8.0f;
    unsigned *dA, *dB; float *dC, *dO;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dO, 2 * 32 * 64 * 16);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dO, 0, 2 * 32 * 64 * 16);
    hipLaunchKernelGGL(probe, dim3(2), dim3(64), 0, 0, (const u32x4 *)dA, (const u32x4 *)dB, (const f32x4 *)dC, (f32x4 *)dO);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> O(2 * 32 * 64 * 4);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
    const char *dn[5] = {"disjoint", "== srcA", "== srcB", "upper half over srcA", "upper half over srcB"};
    int bad_total = 0;
    for (int t = 0; t < 2; ++t) {
        const float *ref0 = &O[((t * 32 + 0) * 64) * 4], *refc = &O[((t * 32 + 5) * 64) * 4];
        for (int k = 0; k < 11; ++k) {
            const float *g = &O[((t * 32 + k) * 64) * 4];
            const float *ref = k < 5 ? ref0 : refc;
            int bad = 0;
            for (int i = 0; i < 256; ++i) bad += g[i] != ref[i];
            printf("%-5s dst %-22s srcC %-12s : %s (%d of 256 values differ)\n", t ? "bf16" : "f16", k < 10 ? dn[k % 5] : "== srcC (in place)",
                   k < 5 ? "constant 0" : (k < 10 ? "own register" : "= dst"), bad ? "WRONG" : "ok", bad);
            bad_total += bad != 0;
        }
    }
    // sanity of the references themselves: C = 0 result + C == own-register result
    for (int t = 0; t < 2; ++t) {
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += O[((t * 32 + 0) * 64) * 4 + i] + C[i] != O[((t * 32 + 5) * 64) * 4 + i];
        printf("%-5s reference check (D(C=0) + C == D(C)): %s\n", t ? "bf16" : "f16", bad ? "MISMATCH" : "ok");
    }
    printf("%d of 22 cases wrong\n", bad_total);
    return 0;
}
