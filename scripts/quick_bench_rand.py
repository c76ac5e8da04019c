"""Ad-hoc timing of the Serra09 chain on i.i.d. random tracks.  usage: quick_bench_rand.py [n_tracks] [T] [exact|f16x2]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 164
T = int(sys.argv[2]) if len(sys.argv) > 2 else 450
P = _lib.serra09_params(arith=sys.argv[3]) if len(sys.argv) > 3 else _lib.serra09_params()
d = synth.rand_set(n, T=T, seed=1234)
ctx = _lib.Context(0)
ctx.upload_pool(d["frames"], d["offsets"])
i, j = np.triu_indices(n, 1)
pairs = np.stack([i, j], 1).astype(np.int32)
ctx.serra09_pairs(pairs[:64], P)
ctx.profile_enable(True)
for rep in range(2):
    ctx.profile_reset()
    t0 = time.time()
    ctx.serra09_pairs(pairs, P)
    dt = time.time() - t0
    print("n=%d T=%d pairs=%d  %.3f s  %.1f pairs/s  %.1f Gcells/s" % (n, T, len(pairs), dt, len(pairs) / dt, len(pairs) * float(T - 9) ** 2 / dt / 1e9))
    for k, v in ctx.profile().items():
        if v["launches"]:
            print("   %-18s %9.3f ms  %3d launches" % (k, v["ms"], v["launches"]))
