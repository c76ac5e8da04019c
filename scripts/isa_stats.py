"""Development aid: static instruction mix of the kernels in a hipcc -S listing (VALU / SALU / MFMA / LDS / memory, spills).
usage: isa_stats.py file.s [name-substring ...]      (hipcc ... --cuda-device-only -S -o file.s x.hip)"""
import re
import sys
from collections import Counter

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
name, body = None, {}
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name = m.group(1)
        body[name] = []
        continue
    if name is None:
        continue
    t = ln.strip()
    if t.startswith(".Lfunc_end"):
        name = None
        continue
    if not t or t[0] in ".;/" or t.endswith(":"):
        continue
    body[name].append(t)
for name, lines in body.items():
    if want and not any(w in name for w in want):
        continue
    c = Counter()
    for t in lines:
        op = t.split()[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("scratch_"):
            c["scratch"] += 1
        else:
            c["mem"] += 1
    sc = [i for i, t in enumerate(lines) if t.startswith("scratch_")]
    print("%s\n   total %d  %s" % (name, len(lines), dict(c)))
    if sc:
        print("   scratch at", sc)
