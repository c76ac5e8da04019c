"""Build-time ISA lint of libacx.so's gfx950 code objects (run by __graft_entry__.build() and tests/test_isa_lint.py; CPU only).

What it guards.  Round 4 met garbage out of v_mfma_f32_16x16x32_f16 in compiled kernels -- a v_mfma_f32_16x16x16_f16 chained
behind it, and a destination the register allocator had placed over the second source operand "returning that operand's bits" --
and worked around both (one opcode per chain, operands kept alive).  Round 6 measured what the hardware actually requires, with
every register FIXED in inline asm and the distance between producer and consumer counted in wait states:

  scripts/ubench/mfma_overlap_probe.hip    a destination over srcA / srcB (whole or half, srcC a constant, a register of its
                                           own or the destination) is HARMLESS, f16 and bf16 alike: 0 of 22 cases wrong
                                           (profiles/r06_mfma_overlap_probe.txt, r06_mfma_waitstate_probe.txt)
  scripts/ubench/mfma_waitstate_probe.hip  the result of a 16-bit-input K = 32 MFMA (v_mfma_f32_16x16x32_f16 / _bf16) is ready for
        the same opcode accumulating in place (srcC = vDst = the result)          after 0 wait states
        v_mfma_f32_16x16x4_f32 taking it as srcC                                  after 0
        any MFMA overwriting it (no read)                                         after 0
        v_mfma_f32_16x16x16_f16 (another 16-bit shape) taking it as srcC          after 5      <- round 4's garbage: stale registers
        an MFMA taking it as srcA / srcB                                          after 7
        a VALU / LDS / memory instruction reading it                              after 7
        a VALU instruction overwriting it                                         after 4
  -- the numbers of a 4-pass XDL operation.  Stale destination registers read too early are what "returns the operand's bits" was:
  the destination had been allocated over the dead operand, and the consumer saw the registers before the MFMA wrote them.

hipcc's hazard recogniser inserts these distances (s_nop / independent instructions); whether it does so for gfx950's double-rate
opcodes in every schedule is what this lint checks on the SHIPPED code objects: for every v_mfma_f32_{16x16x32,32x32x16}_{f16,bf16} it
walks the following instructions of the basic block, counts wait states (an instruction = 1, s_nop N = N + 1) and fails on a
consumer that comes earlier than measured.  Consumers the probe did not measure (another opcode as srcC, a partial srcC overlap)
are held to the strictest measured figure of their kind.  Not covered: consumers behind a taken branch or a loop back-edge (the
walk stops at the end of the block).

A second rule (end of round 6; scripts/ubench/cndmask_probe.hip, profiles/r06_cndmask_probe.txt): of two ADJACENT v_cndmask_b32 ..., vcc
the second costs ~10 cycles instead of 4, every further one ~19 (4 each with the mask in an SGPR pair, or with any instruction between
them) -- hipcc emits the pair for every 64-bit select on a fresh comparison.  lint_selects() reports the runs per kernel and fails on
  * a run of THREE or more in one of the path's hot kernels (HOT_KERNELS), and
  * any run inside SiMPle's sweep (between the first and the last wave_shr DPP move of a simple_kernel -- the recurrence's neighbour
    value, once per step: the running minimum was such a pair there, 7 % of the kernel, until it became one v_min_f64).

usage: python scripts/isa_lint.py [path/to/libacx.so | listing.s ...]     exit status 1 when a violation is found
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("ACX_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PRODUCER = re.compile(r"^v_mfma_f32_(?:16x16x32|32x32x16)_(?:f16|bf16)$")
REGTOK = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
LABEL = re.compile(r"^(?:[0-9a-fA-F]+\s+)?<?([A-Za-z_.$][\w.$]*)>?:\s*$")
# measured wait states (scripts/ubench/mfma_waitstate_probe.hip)
WS_SRCC_OTHER, WS_SRCAB, WS_READ, WS_WRITE = 5, 7, 7, 4
SRCC_FREE = {"v_mfma_f32_16x16x4_f32"}          # measured: takes the result as srcC back to back
HORIZON = max(WS_SRCC_OTHER, WS_SRCAB, WS_READ, WS_WRITE)
# non-MFMA instructions whose FIRST operand is only written (everything else of theirs, and every operand of the rest, is a read)
WRITES_FIRST = re.compile(r"^(v_(?!cmp|cmpx|fmac|mac|dot\d|pk_fmac|swap|readlane|readfirstlane|mfma|smfmac)|ds_read|ds_bpermute|ds_permute|ds_swizzle|"
                          r"global_load|buffer_load|flat_load|scratch_load|v_accvgpr_read|v_accvgpr_write)")
BLOCK_END = re.compile(r"^(s_endpgm|s_branch|s_setpc_b64|s_swappc_b64|s_trap)")


def code_objects(lib_path):
    """The gfx950 code objects inside a host shared library / object (one clang offload bundle per translation unit in .hip_fatbin)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        data = open(fat, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        o = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tsize = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tsize].decode()
            o += tsize
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(MAGIC, pos + 1)
    return out


def disassemble(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(elf_bytes)
        f.flush()
        return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], check=True,
                              stdout=subprocess.PIPE, universal_newlines=True).stdout


def _regs(text):
    """set of ('v' | 'a', n) named in an operand string"""
    out = set()
    for m in REGTOK.finditer(text):
        if m.group(1):
            out.update((m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def _parse(line):
    """(opcode, [operand strings]) of an instruction line, None for labels / directives / blanks."""
    t = line.split("//")[0].split(";")[0].strip()
    if not t or t.endswith(":") or t[0] in ".<" or not re.match(r"^[a-z]", t):
        return None
    parts = t.split(None, 1)
    return parts[0], ([o.strip() for o in parts[1].split(",")] if len(parts) > 1 else [])


def lint_text(text):
    """(checked, violations, stats) of a disassembly / -S listing; violations = [{kernel, producer, consumer, wait_states, needs, reason}]."""
    lines = text.split("\n")
    insts = []                                    # (kernel, opcode, operands) or None at a block boundary
    kernel = "?"
    for ln in lines:
        if not ln.startswith(("\t", " ")):
            lab = LABEL.match(ln.split(";")[0].strip())
            if lab:
                if not lab.group(1).startswith((".L", "BB")):
                    kernel = lab.group(1)
                insts.append(None)                # a label: something may jump here
                continue
        p = _parse(ln)
        if p:
            insts.append((kernel, p[0], p[1]))
    checked, bad = 0, []
    stats = {"dst_over_srcA_or_srcB_harmless": 0}
    for i, ins in enumerate(insts):
        if ins is None or not PRODUCER.match(ins[1]) or len(ins[2]) < 4:
            continue
        kern, pop, pops = ins
        D = _regs(pops[0])
        checked += 1
        if D & (_regs(pops[1]) | _regs(pops[2])):
            stats["dst_over_srcA_or_srcB_harmless"] += 1
        ws = 0
        for j in range(i + 1, len(insts)):
            c = insts[j]
            if c is None or ws >= HORIZON:
                break
            _, cop, cops = c
            need, why = 0, None
            if cop.startswith(("v_mfma", "v_smfmac")) and len(cops) >= 4:
                cd, ca, cb, cc = (_regs(o) for o in cops[:4])
                if D & (ca | cb):
                    need, why = WS_SRCAB, "an MFMA reads the result as srcA / srcB"
                elif D & cc:
                    if cop == pop and cc == D:
                        need = 0                  # same opcode, srcC = the whole result: forwarded
                    elif cop in SRCC_FREE and cc == D:
                        need = 0
                    else:
                        need, why = WS_SRCC_OTHER, "an MFMA of another opcode (or over part of the registers) takes the result as srcC"
            elif not cop.startswith("s_"):
                first_written = bool(WRITES_FIRST.match(cop)) and cops
                reads = set()
                for k, o in enumerate(cops):
                    if not (k == 0 and first_written):
                        reads |= _regs(o)
                if D & reads:
                    need, why = WS_READ, "the result is read by a non-MFMA instruction"
                elif first_written and D & _regs(cops[0]):
                    need, why = WS_WRITE, "a non-MFMA instruction overwrites the destination"
            if need > ws:
                bad.append({"kernel": kern, "producer": pop + " " + ", ".join(pops), "consumer": cop + " " + ", ".join(cops),
                            "wait_states": ws, "needs": need, "reason": why})
            if BLOCK_END.match(cop):
                break
            if cop == "s_nop" and cops:
                try:
                    ws += int(cops[0], 0) + 1
                except ValueError:
                    ws += 1
            else:
                ws += 1
    return checked, bad, stats


HOT_KERNELS = re.compile(r"band_kernel|band2_kernel|qmax_bits_h16|sw_bits_h16|ef_rowstat|ef_colstat|ef_gemm_rect|simple_kernel")


def lint_selects(text):
    """(kernels seen, {kernel: {run length: count}}, violations) for runs of adjacent v_cndmask_b32 ..., vcc in a disassembly / -S listing."""
    kernel, run, start = "?", 0, 0
    runs, seen, bad = {}, set(), []
    vmin = {}                                     # simple_kernel: instruction indices of its wave_shr DPP moves (the sweep lies between them)
    pend = []                                     # (kernel, first instruction index, length)
    n = 0

    def close():
        nonlocal run
        if run >= 2:
            runs.setdefault(kernel, {})
            runs[kernel][run] = runs[kernel].get(run, 0) + 1
            pend.append((kernel, start, run))
        run = 0

    for ln in text.split("\n"):
        if not ln.startswith(("\t", " ")):
            lab = LABEL.match(ln.split(";")[0].strip())
            if lab:
                close()
                if not lab.group(1).startswith((".L", "BB")):
                    kernel = lab.group(1)
                    seen.add(kernel)
                continue
        p = _parse(ln)
        if not p:
            continue
        n += 1
        if p[0].startswith("v_mov_b32_dpp") and "wave_shr" in ln and "simple_kernel" in kernel:
            vmin.setdefault(kernel, []).append(n)
        if p[0].startswith("v_cndmask_b32") and p[1] and p[1][-1] == "vcc":
            if run == 0:
                start = n
            run += 1
        else:
            close()
    close()
    for kern, at, length in pend:
        if not HOT_KERNELS.search(kern):
            continue
        if length >= 3:
            bad.append({"kernel": kern, "run": length, "reason": "three or more adjacent v_cndmask_b32 ..., vcc in a hot kernel (~19 cycles each from the third on)"})
        elif "simple_kernel" in kern and len(vmin.get(kern, [])) >= 2 and vmin[kern][0] < at < vmin[kern][-1]:
            bad.append({"kernel": kern, "run": length, "reason": "adjacent v_cndmask_b32 ..., vcc inside SiMPle's sweep (a 64-bit select on a fresh comparison: ~10 extra cycles per step)"})
    return len(seen), runs, bad


def lint_library(lib_path):
    checked, bad, dst_over = 0, [], 0
    sel_bad, sel_runs, kernels = [], 0, 0
    objs = code_objects(lib_path)
    for co in objs:
        dis = disassemble(co)
        n, b, st = lint_text(dis)
        checked += n
        bad += b
        dst_over += st["dst_over_srcA_or_srcB_harmless"]
        k, runs, sb = lint_selects(dis)
        kernels += k
        sel_runs += sum(c for r in runs.values() for c in r.values())
        sel_bad += sb
    return {"library": lib_path, "code_objects": len(objs), "mfma_checked": checked,
            "dst_over_srcA_or_srcB_harmless": dst_over, "violations": bad,
            "kernels": kernels, "adjacent_select_runs": sel_runs, "select_violations": sel_bad}


def main(argv):
    paths = argv or [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "acoss_amd", "csrc", "libacx.so")]
    rc = 0
    for p in paths:
        if p.endswith(".s"):
            txt = open(p).read()
            n, b, st = lint_text(txt)
            k, runs, sb = lint_selects(txt)
            rec = {"listing": p, "mfma_checked": n, "violations": b, "kernels": k,
                   "adjacent_select_runs": sum(c for r in runs.values() for c in r.values()), "select_violations": sb}
            rec.update(st)
        else:
            rec = lint_library(p)
        print(json.dumps(rec, indent=1))
        if rec["violations"] or rec["select_violations"] or not rec["mfma_checked"]:
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
