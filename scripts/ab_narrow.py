"""Development aid: the Serra09 chain on the short-row workloads (i.i.d. T = 450 / 250, covers80-shaped 150-650), several repetitions,
best and median rate + the band kernels' share.  Pick the build with ACX_LIB=build_ab/libacx_<name>.so / ACX_BAND2=0|1.
usage: ab_narrow.py [reps] [prof]      (prof: also read the library's per-kernel event clocks -- they cost 10-20 % of a short call)"""
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
prof = len(sys.argv) > 2
ctx = _lib.Context(0)


def run(label, d):
    n = len(d["offsets"]) - 1
    ctx.upload_pool(d["frames"], d["offsets"])
    i, j = np.triu_indices(n, 1)
    pairs = np.stack([i, j], 1).astype(np.int32)
    L = (np.diff(d["offsets"]) - 9).astype(np.float64)
    cells = float(np.sum(L[i] * L[j]))
    ctx.serra09_pairs(pairs[:64])
    ctx.profile_enable(prof)
    ts, band = [], []
    for _ in range(reps):
        ctx.profile_reset()
        t0 = time.time()
        out = ctx.serra09_pairs(pairs)
        ts.append(time.time() - t0)
        pr = ctx.profile() if prof else {}
        band.append(sum(v["ms"] for k, v in pr.items() if "band" in k))
    ts = np.array(ts)
    print("%-22s pairs=%d  best %.1f / median %.1f Gcells/s  (%.0f k pairs/s best)  band kernels %.3f ms (best)  crc %08x" % (
        label, len(pairs), cells / ts.min() / 1e9, cells / np.median(ts) / 1e9, len(pairs) / ts.min() / 1e3, min(band),
        zlib.crc32(np.ascontiguousarray(out, dtype=np.float32).tobytes())), flush=True)


run("iid T=450 n=164", synth.rand_set(164, T=450, seed=1234))
run("iid T=250 n=200", synth.rand_set(200, T=250, seed=1234))
run("covers 150-650 n=164", synth.cover_set(clique_sizes=[2] * 82, seed=4321, t_range=(150, 650)))
run("covers 300-600 n=164", synth.cover_set(clique_sizes=[2] * 77 + [3, 3, 4], seed=4321, t_range=(300, 600)))
