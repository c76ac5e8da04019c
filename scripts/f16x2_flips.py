"""Serra09: the fraction of recurrence cells that change side of a threshold under the opt-in f16x2 Gram (debug pairs,
both arithmetics), per track length (profiles/r04_f16x2.md).  (python scripts/f16x2_flips.py, on a GPU)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib, synth
import oracle
d = synth.covers80_shaped(seed=100, t_range=(150, 650))
ctx = _lib.Context(0)
ctx.upload_pool(d["frames"], d["offsets"])
tot = flips = 0
dd = []
for (i, j) in ((0, 1), (0, 2), (5, 9), (10, 11), (20, 33), (40, 41), (7, 90), (100, 101)):
    e = ctx.serra09_debug_pair(i, j, _lib.serra09_params())
    f = ctx.serra09_debug_pair(i, j, _lib.serra09_params(arith="f16x2"))
    Re = (e["d2"] <= e["thr_q"][:, None]) & (e["d2"] <= e["thr_r"][None, :])
    Rf = (f["d2"] <= f["thr_q"][:, None]) & (f["d2"] <= f["thr_r"][None, :])
    q = d["frames"][d["offsets"][i]:d["offsets"][i + 1]]; r = d["frames"][d["offsets"][j]:d["offsets"][j + 1]]
    so, it = oracle.serra09_pair(q, r, oracle.serra09_params(arith="seq108"), want_intermediates=True)
    tot += Re.size; flips += int(np.sum(Re != Rf))
    dif = np.abs(e["d2"].astype(np.float64) - f["d2"])
    dif_s = np.abs(e["d2"].astype(np.float64) - it["d"].astype(np.float64) ** 2)
    print("pair (%d,%d) %s: flips f16x2 %d, seq108 %d | |d2 diff| f16x2: max %.2e mean %.2e, nonzero %.3f | seq108 (d^2 of rooted): mean %.2e | score %.1f / %.1f / %.1f" % (
        i, j, Re.shape, int(np.sum(Re != Rf)), int(np.sum(Re.astype(np.uint8) != it["R"])), dif.max(), dif.mean(), np.mean(dif > 0), dif_s.mean(), e["score"], f["score"], so))
print("flipped-cell fraction f16x2 vs exact: %.3e (%d of %d)" % (flips / tot, flips, tot))
