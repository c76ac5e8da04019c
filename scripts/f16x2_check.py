"""Serra09, exact chain vs the opt-in f16x2 Gram on i.i.d. tracks of one length: throughput and score differences.
(python scripts/f16x2_check.py [n_tracks] [frames], on a GPU)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib, synth
import oracle
for name, d in (("covers 150-650", synth.covers80_shaped(seed=100, t_range=(150, 650))), ("covers 500-1000", synth.cover_set(clique_sizes=[2] * 40, seed=5, t_range=(500, 1000))),
                ("rand T=2000", synth.rand_set(48, T=2000, seed=1234))):
    n = len(d["offsets"]) - 1
    ctx = _lib.Context(0)
    ctx.upload_pool(d["frames"], d["offsets"])
    pairs = oracle.all_pairs(n, True).astype(np.int32)
    pe, pf = _lib.serra09_params(), _lib.serra09_params(arith="f16x2")
    ctx.serra09_pairs(pairs[:64], pe); ctx.serra09_pairs(pairs[:64], pf)
    t0 = time.time(); se = ctx.serra09_pairs(pairs, pe); te = time.time() - t0
    t0 = time.time(); sf = ctx.serra09_pairs(pairs, pf); tf = time.time() - t0
    diff = np.abs(se - sf)
    print("%-16s %6d pairs: exact %.1f k pairs/s, f16x2 %.1f k pairs/s (x %.3f); identical %.4f, max |diff| %.1f, >2.0: %d" % (
        name, len(pairs), len(pairs) / te / 1e3, len(pairs) / tf / 1e3, te / tf, np.mean(diff == 0), diff.max(), int(np.sum(diff > 2.0))))
    ctx.close()
