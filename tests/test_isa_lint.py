"""
CPU suite: the build-time ISA lint (scripts/isa_lint.py) -- the measured wait-state requirements of gfx950's 16-bit-input
K = 32 MFMAs, checked on the code objects that ship.

  * the shipped libacx.so is green, and every such MFMA in it was looked at;
  * each measured rule is red one wait state too early and green at the measured distance (hand-written listings);
  * a real code object with a too-early consumer (inline asm, compiled here by hipcc) is red: the path the build takes --
    fat binary -> code object -> disassembly -> rule -- finds what it is there to find;
  * a destination over srcA / srcB, which round 4 suspected and scripts/ubench/mfma_overlap_probe.hip cleared, is counted, not failed;
  * the second rule -- adjacent v_cndmask_b32 ..., vcc (scripts/ubench/cndmask_probe.hip): three in a row in a hot kernel, or a pair
    inside SiMPle's sweep, is red; the forms that cost nothing are green.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _lint():
    import isa_lint
    return isa_lint


def test_shipped_library_is_green():
    lib = os.path.join(ROOT, "acoss_amd", "csrc", "libacx.so")
    if not os.path.exists(lib):
        pytest.skip("libacx.so not built")
    rec = _lint().lint_library(lib)
    assert rec["code_objects"] == 2 and rec["mfma_checked"] >= 500, rec
    assert rec["violations"] == [], rec["violations"][:5]
    assert rec["select_violations"] == [] and rec["kernels"] > 100, rec["select_violations"][:5]


P = "\tv_mfma_f32_16x16x32_f16 v[56:59], v[40:43], v[44:47], 0\n"


def _listing(consumer, ws):
    nop = "" if ws == 0 else "\ts_nop %d\n" % (ws - 1)
    return "kern:\n" + P + nop + "\t" + consumer + "\n\ts_endpgm\n"


@pytest.mark.parametrize("consumer,need", [
    ("v_mfma_f32_16x16x32_f16 v[56:59], v[60:63], v[64:67], v[56:59]", 0),          # same opcode, in place
    ("v_mfma_f32_16x16x4_f32 v[56:59], v60, v64, v[56:59]", 0),
    ("v_mfma_f32_16x16x16_f16 v[56:59], v[60:61], v[64:65], 0", 0),                # overwrites, no read
    ("v_mfma_f32_16x16x16_f16 v[56:59], v[60:61], v[64:65], v[56:59]", 5),         # round 4's chain
    ("v_mfma_f32_16x16x32_bf16 v[56:59], v[60:63], v[64:67], v[56:59]", 5),        # (unmeasured opcode pair: the strict figure)
    ("v_mfma_f32_16x16x32_f16 v[56:59], v[60:63], v[64:67], v[58:61]", 5),         # partial srcC overlap
    ("v_mfma_f32_16x16x32_f16 v[68:71], v[56:59], v[64:67], 0", 7),
    ("v_mfma_f32_16x16x32_f16 v[68:71], v[60:63], v[58:61], 0", 7),
    ("v_add_f32_e32 v68, v56, v56", 7),
    ("global_store_dword v1, v57, s[10:11] offset:64", 7),
    ("ds_write_b128 v2, v[56:59]", 7),
    ("v_fmac_f32_e32 v56, v1, v2", 7),                                             # reads its destination
    ("v_mov_b32_e32 v58, 1.0", 4),
    ("ds_read_b128 v[56:59], v2", 4),
])
def test_every_measured_rule(consumer, need):
    lint = _lint()
    n, bad, _ = lint.lint_text(_listing(consumer, need))
    assert n >= 1 and bad == [], bad
    if need > 0:
        n, bad, _ = lint.lint_text(_listing(consumer, need - 1))
        assert len(bad) == 1 and bad[0]["needs"] == need and bad[0]["wait_states"] == need - 1, bad
    # independent instructions count as wait states, a label ends the walk
    filler = "".join("\tv_mov_b32_e32 v%d, 0\n" % (100 + k) for k in range(need))
    n, bad, _ = lint.lint_text("kern:\n" + P + filler + "\t" + consumer + "\n\ts_endpgm\n")
    assert bad == []
    n, bad, _ = lint.lint_text("kern:\n" + P + ".LBB0_1:\n\t" + consumer + "\n\ts_endpgm\n")
    assert bad == []


def test_destination_over_a_source_is_counted_not_failed():
    n, bad, st = _lint().lint_text("kern:\n\tv_mfma_f32_16x16x32_f16 v[44:47], v[40:43], v[44:47], 0\n\ts_nop 7\n\tv_mov_b32_e32 v1, v44\n")
    assert n == 1 and bad == [] and st["dst_over_srcA_or_srcB_harmless"] == 1


NEGATIVE = r'''
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void too_early(f32x4 *out, const f32x4 *in)
{
    f32x4 o;
    const f32x4 a = in[threadIdx.x], b = in[64 + threadIdx.x];
    asm volatile("v_mov_b32 v40, %[a0]\n\tv_mov_b32 v41, %[a1]\n\tv_mov_b32 v42, %[a2]\n\tv_mov_b32 v43, %[a3]\n\t"
                 "v_mov_b32 v44, %[b0]\n\tv_mov_b32 v45, %[b1]\n\tv_mov_b32 v46, %[b2]\n\tv_mov_b32 v47, %[b3]\n\t"
                 "s_nop 7\n\t"
                 "v_mfma_f32_16x16x32_f16 v[56:59], v[40:43], v[44:47], 0\n\t"
                 "s_nop %[W]\n\t"
                 "v_mfma_f32_16x16x16_f16 v[56:59], v[40:41], v[44:45], v[56:59]\n\t"
                 "s_nop 15\n\t"
                 "v_mov_b32 %[o0], v56\n\tv_mov_b32 %[o1], v57\n\tv_mov_b32 %[o2], v58\n\tv_mov_b32 %[o3], v59\n\t"
                 : [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3])
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]),
                   [W] "n"(ACX_NEG_NOP)
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v56", "v57", "v58", "v59");
    out[threadIdx.x] = o;
}
'''


@pytest.mark.timeout(600)
@pytest.mark.parametrize("nop,red", [(2, True), (4, False)])       # s_nop 2 = 3 wait states (too early), s_nop 4 = 5 (the measured distance)
def test_a_compiled_code_object_with_a_too_early_consumer(tmp_path, nop, red):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "neg.hip"
    src.write_text(NEGATIVE)
    so = tmp_path / "libneg.so"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-w", "-fPIC", "-shared", "-DACX_NEG_NOP=%d" % nop, "-o", str(so), str(src)])
    rec = _lint().lint_library(str(so))
    assert rec["code_objects"] == 1 and rec["mfma_checked"] == 1, rec
    if red:
        assert len(rec["violations"]) == 1 and rec["violations"][0]["needs"] == 5 and rec["violations"][0]["wait_states"] == 3, rec
    else:
        assert rec["violations"] == [], rec


SELECT = "\tv_cndmask_b32_e32 v%d, v%d, v%d, vcc\n"
DPP = "\tv_mov_b32_dpp v20, v70 wave_shr:1 row_mask:0xf bank_mask:0xf\n"


@pytest.mark.parametrize("kernel,body,n_bad", [
    # three adjacent selects on vcc in a hot kernel: ~19 cycles for the third (profiles/r06_cndmask_probe.txt)
    ("_ZN3acx12band2_kernelILi9ELi0ELb0ELi16ELi32EEEvPKf", "\tv_cmp_lt_f32_e32 vcc, v0, v1\n" + SELECT % (2, 2, 3) + SELECT % (4, 4, 5) + SELECT % (6, 6, 7), 1),
    # ... a pair there is tolerated, and so are three with the mask in an SGPR pair or with an instruction between them
    ("_ZN3acx12band2_kernelILi9ELi0ELb0ELi16ELi32EEEvPKf", "\tv_cmp_lt_f32_e32 vcc, v0, v1\n" + SELECT % (2, 2, 3) + SELECT % (4, 4, 5), 0),
    ("_ZN3acx12band2_kernelILi9ELi0ELb0ELi16ELi32EEEvPKf", "\tv_cndmask_b32_e64 v2, v2, v3, s[2:3]\n" * 3, 0),
    ("_ZN3acx12band2_kernelILi9ELi0ELb0ELi16ELi32EEEvPKf", (SELECT % (2, 2, 3) + "\tv_add_f32_e32 v9, v9, v8\n") * 3, 0),
    # three in a kernel that is not on the path's hot list: counted, not failed
    ("_ZN3acx14normtab_kernelILi6EEEvPKf", SELECT % (2, 2, 3) + SELECT % (4, 4, 5) + SELECT % (6, 6, 7), 0),
    # SiMPle: a pair inside the sweep (between its wave_shr DPP moves) is what round 6 removed; outside the sweep it is tolerated
    ("_ZN3acx13simple_kernelILi10EEEvPKd", DPP + "\tv_cmp_lt_f64_e32 vcc, v[20:21], v[62:63]\n" + SELECT % (63, 63, 21) + SELECT % (62, 62, 20) + DPP, 1),
    ("_ZN3acx13simple_kernelILi10EEEvPKd", SELECT % (63, 63, 21) + SELECT % (62, 62, 20) + DPP + "\tv_min_f64 v[68:69], v[6:7], v[68:69]\n" + DPP, 0),
])
def test_adjacent_selects_on_vcc(kernel, body, n_bad):
    L = _lint()
    k, runs, bad = L.lint_selects(kernel + ":\n" + body + "\ts_endpgm\n")
    assert k == 1 and len(bad) == n_bad, (runs, bad)
