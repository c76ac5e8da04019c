"""Generates scripts/ubench/valu_table.hip: the issue cost of single gfx950 vector instructions (cycles per wave-instruction and
SIMD at 1 / 2 / 4 / 8 resident waves per SIMD), one kernel per instruction, eight independent register chains each.
A development aid (profiles/r06_valu_table.txt is its output on one MI355X):   python scripts/ubench/gen_valu_table.py && \
hipcc -O3 --offload-arch=gfx950 -o scripts/ubench/valu_table scripts/ubench/valu_table.hip"""
import os

# (name, asm template for register i; {r} = the chain's register, {b}/{c} = loop-invariant operands, kind: f = float regs, u = unsigned, d = double)
OPS = [
    ("v_add_f32", "v_add_f32 {r}, {r}, {b}", "f"), ("v_sub_f32", "v_sub_f32 {r}, {r}, {b}", "f"), ("v_mul_f32", "v_mul_f32 {r}, {r}, {b}", "f"),
    ("v_fma_f32", "v_fma_f32 {r}, {r}, {b}, {c}", "f"), ("v_fmac_f32", "v_fmac_f32 {r}, {b}, {c}", "f"),
    ("v_fma_f32 (inline constant + SGPR: the sweep's -2 xy + |x|^2)", "v_fma_f32 {r}, {r}, -2.0, s22", "f"),
    ("v_add_f32 (an SGPR term)", "v_add_f32 {r}, s22, {r}", "f"), ("v_mul_f32 (a 32-bit literal)", "v_mul_f32 {r}, 0x3fb8aa3b, {r}", "f"),
    ("v_max_f32 (inline 0: the clamp)", "v_max_f32 {r}, 0, {r}", "f"), ("v_add_u32 (a 32-bit literal)", "v_add_u32 {r}, 0x12345, {r}", "u"),
    ("v_min_f32", "v_min_f32 {r}, {r}, {b}", "f"), ("v_max_f32", "v_max_f32 {r}, {r}, {b}", "f"),
    ("v_med3_f32", "v_med3_f32 {r}, {r}, {b}, {c}", "f"), ("v_min3_f32", "v_min3_f32 {r}, {r}, {b}, {c}", "f"),
    ("v_exp_f32", "v_exp_f32 {r}, {r}", "f"), ("v_rcp_f32", "v_rcp_f32 {r}, {r}", "f"), ("v_sqrt_f32", "v_sqrt_f32 {r}, {r}", "f"),
    ("v_floor_f32", "v_floor_f32 {r}, {r}", "f"), ("v_cvt_u32_f32", "v_cvt_u32_f32 {r}, {r}", "f"), ("v_cvt_f32_u32", "v_cvt_f32_u32 {r}, {r}", "f"),
    ("v_mov_b32", "v_mov_b32 {r}, {b}", "f"),
    ("v_cmp_lt_f32 (vcc)", "v_cmp_lt_f32 vcc, {r}, {b}", "f"), ("v_cmp_lt_f32 (sgpr pair)", "v_cmp_lt_f32 s[20:21], {r}, {b}", "f"),
    ("v_cndmask_b32", "v_cndmask_b32 {r}, {r}, {b}, vcc", "f"),
    ("v_cndmask_b32 (sgpr pair mask)", "v_cndmask_b32 {r}, {r}, {b}, s[22:23]", "f"),
    ("v_cndmask_b32 (own value in src1)", "v_cndmask_b32 {r}, {b}, {r}, vcc", "f"),
    ("v_cmp_lt_f32 + v_cndmask_b32 (per instruction)", "v_cmp_lt_f32 vcc, {r}, {b}\\n v_cndmask_b32 {r}, {r}, {c}, vcc", "f2"),
    ("v_cmp_class_f32", "v_cmp_class_f32 vcc, {r}, {b}", "f"),
    ("v_add_u32", "v_add_u32 {r}, {r}, {b}", "u"), ("v_sub_u32", "v_sub_u32 {r}, {r}, {b}", "u"), ("v_addc_co_u32", "v_addc_co_u32 {r}, vcc, {r}, {b}, vcc", "u"),
    ("v_and_b32", "v_and_b32 {r}, {r}, {b}", "u"), ("v_or_b32", "v_or_b32 {r}, {r}, {b}", "u"), ("v_xor_b32", "v_xor_b32 {r}, {r}, {b}", "u"),
    ("v_not_b32", "v_not_b32 {r}, {r}", "u"),
    ("v_lshlrev_b32", "v_lshlrev_b32 {r}, 1, {r}", "u"), ("v_lshlrev_b32 (shift in a register)", "v_lshlrev_b32 {r}, {c}, {r}", "u"), ("v_lshrrev_b32", "v_lshrrev_b32 {r}, 1, {r}", "u"),
    ("v_lshrrev_b32 (shift in a register)", "v_lshrrev_b32 {r}, {c}, {r}", "u"), ("v_ashrrev_i32", "v_ashrrev_i32 {r}, 1, {r}", "u"),
    ("v_min_u32", "v_min_u32 {r}, {r}, {b}", "u"), ("v_max_u32", "v_max_u32 {r}, {r}, {b}", "u"), ("v_min_i32", "v_min_i32 {r}, {r}, {b}", "u"),
    ("v_cmp_lt_u32 (vcc)", "v_cmp_lt_u32 vcc, {r}, {b}", "u"), ("v_cmp_eq_u32 (vcc)", "v_cmp_eq_u32 vcc, {r}, {b}", "u"),
    ("v_mul_u32_u24", "v_mul_u32_u24 {r}, {r}, {b}", "u"), ("v_mad_u32_u24", "v_mad_u32_u24 {r}, {r}, {b}, {c}", "u"), ("v_mul_lo_u32", "v_mul_lo_u32 {r}, {r}, {b}", "u"),
    ("v_add3_u32", "v_add3_u32 {r}, {r}, {b}, {c}", "u"), ("v_lshl_add_u32", "v_lshl_add_u32 {r}, {r}, 2, {b}", "u"), ("v_add_lshl_u32", "v_add_lshl_u32 {r}, {r}, {b}, 2", "u"),
    ("v_and_or_b32", "v_and_or_b32 {r}, {r}, {b}, {c}", "u"), ("v_lshl_or_b32", "v_lshl_or_b32 {r}, {r}, 2, {b}", "u"), ("v_or3_b32", "v_or3_b32 {r}, {r}, {b}, {c}", "u"),
    ("v_bfe_u32", "v_bfe_u32 {r}, {r}, 3, 5", "u"), ("v_bfi_b32", "v_bfi_b32 {r}, {b}, {r}, {c}", "u"), ("v_perm_b32", "v_perm_b32 {r}, {r}, {b}, {c}", "u"),
    ("v_alignbit_b32", "v_alignbit_b32 {r}, {r}, {b}, 7", "u"), ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {r}, {r}, {b}", "u"), ("v_ffbh_u32", "v_ffbh_u32 {r}, {r}", "u"),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 {r}, {b}, {r}", "u"),
    ("v_pk_add_u16", "v_pk_add_u16 {r}, {r}, {b}", "u"), ("v_pk_sub_i16", "v_pk_sub_i16 {r}, {r}, {b}", "u"), ("v_pk_max_i16", "v_pk_max_i16 {r}, {r}, {b}", "u"),
    ("v_pk_min_u16", "v_pk_min_u16 {r}, {r}, {b}", "u"), ("v_pk_lshlrev_b16", "v_pk_lshlrev_b16 {r}, 1, {r}", "u"), ("v_pk_mul_lo_u16", "v_pk_mul_lo_u16 {r}, {r}, {b}", "u"),
    ("v_pk_add_f16", "v_pk_add_f16 {r}, {r}, {b}", "u"), ("v_pk_fma_f16", "v_pk_fma_f16 {r}, {r}, {b}, {c}", "u"),
    ("v_mov_b32_dpp row_shr:1", "v_mov_b32_dpp {r}, {r} row_shr:1 row_mask:0xf bank_mask:0xf", "u"),
    ("v_add_u32_dpp row_shr:1", "v_add_u32_dpp {r}, {r}, {r} row_shr:1 row_mask:0xf bank_mask:0xf", "u"),
    ("v_min_u32_dpp row_ror:4", "v_min_u32_dpp {r}, {r}, {r} row_ror:4 row_mask:0xf bank_mask:0xf", "u"),
    ("v_add_f32_dpp quad_perm", "v_add_f32_dpp {r}, {r}, {r} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "f"),
    ("v_mov_b32_dpp row_bcast:15", "v_mov_b32_dpp {r}, {r} row_bcast:15 row_mask:0xa bank_mask:0xf", "u"),
    ("v_add_u32_sdwa", "v_add_u32_sdwa {r}, {r}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD", "u"),
    ("v_readlane_b32", "v_readlane_b32 s20, {r}, 3", "u"), ("v_readfirstlane_b32", "v_readfirstlane_b32 s20, {r}", "u"),
    ("v_permlane16_swap_b32", "v_permlane16_swap_b32 {r}, {r}", "u"), ("v_permlane32_swap_b32", "v_permlane32_swap_b32 {r}, {r}", "u"),
    ("v_pk_add_f32", "v_pk_add_f32 {r}, {r}, {b}", "d"), ("v_pk_fma_f32", "v_pk_fma_f32 {r}, {r}, {b}, {c}", "d"), ("v_pk_mov_b32", "v_pk_mov_b32 {r}, {r}, {b}", "d"),
    ("v_add_f64", "v_add_f64 {r}, {r}, {b}", "d"), ("v_mul_f64", "v_mul_f64 {r}, {r}, {b}", "d"), ("v_fma_f64", "v_fma_f64 {r}, {r}, {b}, {c}", "d"),
    ("v_fmac_f64", "v_fmac_f64 {r}, {b}, {c}", "d"), ("v_fmac_f64 (an SGPR pair as a factor)", "v_fmac_f64 {r}, s[22:23], {c}", "d"),
    ("v_fmac_f64, ONE dependent chain", "v_fmac_f64 %0, {b}, {c}", "d"), ("v_fmac_f32, ONE dependent chain", "v_fmac_f32 %0, {b}, {c}", "f"),
    ("v_add_f64 (an SGPR pair as a term)", "v_add_f64 {r}, {r}, s[22:23]", "d"),
    ("v_min_f64", "v_min_f64 {r}, {r}, {b}", "d"), ("v_cmp_lt_f64 (vcc)", "v_cmp_lt_f64 vcc, {r}, {b}", "d"), ("v_lshlrev_b64", "v_lshlrev_b64 {r}, 1, {r}", "d"),
    ("v_lshl_add_u64", "v_lshl_add_u64 {r}, {r}, 1, {b}", "d"), ("v_mad_u64_u32", "v_mad_u64_u32 {r}, vcc, {b}, {c}, {r}", "dmix"),
    ("s_nop 0 (issue slot only)", "s_nop 0", "f"),
]

HEAD = r'''// GENERATED by scripts/ubench/gen_valu_table.py -- do not edit.  Issue cost of single gfx950 vector instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define REP 8           /* copies of the 8-instruction group per loop iteration */
template <int OP> __global__ __launch_bounds__(256) void k(float *out, int n);
template <int OP> constexpr int per_step() { return 1; }
'''


def kernel(idx, name, tmpl, kind):
    regs = ["%%%d" % i for i in range(8)]
    lines = []
    for i in range(8):
        lines.append(tmpl.format(r=regs[i], b="%8", c="%9"))
    body = "\\n ".join(lines)
    per = 2 if kind == "f2" else 1
    if kind in ("f", "f2"):
        decl = "float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7; float b = 1.0001f, c = 0.5f;"
        fin = "a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7"
    elif kind == "u":
        decl = "unsigned a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7; unsigned b = 0x01010101u * (lane + 1), c = 0x00ff00ffu;"
        fin = "(float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)"
    elif kind == "d":
        decl = "double a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7; double b = 1.0001, c = 0.5;"
        fin = "(float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)"
    else:       # dmix: 64-bit chains, 32-bit invariant operands
        decl = "unsigned long long a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7; unsigned b = 3u + lane, c = 5u;"
        fin = "(float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)"
    return '''template <> constexpr int per_step<%d>() { return %d; }
template <> __global__ __launch_bounds__(256) void k<%d>(float *out, int n)      // %s
{
    const int lane = threadIdx.x & 63;
    %s
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r)
            asm volatile("%s"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");
    }
    out[blockIdx.x * 256 + threadIdx.x] = %s;
}
''' % (idx, per, idx, name, decl, body, fin)


TAIL = r'''
template <int OP> double run(float *d, int wg_per_cu, int ncu)
{
    const int n = ITER, blocks = wg_per_cu * ncu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 8);
    (void)hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)n * REP * 8 * per_step<OP>() * wg_per_cu);      // 256 threads = one wave per SIMD and workgroup
        if (cyc < best) best = cyc;
    }
    return best;
}
template <int OP> void row(const char *name, float *d, int ncu)
{
    printf("| `%s` | %.2f | %.2f | %.2f | %.2f |\n", name, run<OP>(d, 1, ncu), run<OP>(d, 2, ncu), run<OP>(d, 4, ncu), run<OP>(d, 8, ncu));
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float *d; (void)hipMalloc(&d, sizeof(float) * 256 * 8 * ncu);
    printf("%s, %d CUs: cycles (at 2.4 GHz; launch time / instructions, launch overhead included) per wave-instruction and SIMD, eight independent chains per wave\n", p.name, ncu);
    printf("| instruction | 1 wave / SIMD | 2 | 4 | 8 |\n|---|---|---|---|---|\n");
'''


def main():
    src = [HEAD]
    for i, (name, tmpl, kind) in enumerate(OPS):
        src.append(kernel(i, name, tmpl, kind))
    src.append(TAIL)
    for i, (name, _, _) in enumerate(OPS):
        src.append('    row<%d>("%s", d, ncu);\n' % (i, name))
    src.append("    return 0;\n}\n")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_table.hip"), "w") as f:
        f.write("".join(src))


if __name__ == "__main__":
    main()
