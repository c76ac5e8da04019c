"""Serra09, exact chain vs the opt-in f16x2 Gram on three cover-structured sets: pairs/s, fraction of identical scores, |diff| <= 2,
MAP / MR / Top-1 of both (the numbers of profiles/r04_f16x2.md).  (python scripts/f16x2_map.py, on a GPU)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib, synth
import oracle
def cliques(labels):
    cl = {}
    for i, l in enumerate(labels): cl.setdefault(l, []).append(i)
    return list(cl.values())
for name, d in (("covers80-shaped 150-650", synth.covers80_shaped(seed=100, t_range=(150, 650))),
                ("covers80-shaped hard (noise 1.0, halves)", synth.covers80_shaped(seed=7, t_range=(300, 600), noise=1.0, segment_keep=0.5)),
                ("cover set 60 works x 3, 500-1000", synth.cover_set(n_works=60, versions=3, seed=11, t_range=(500, 1000), noise=0.6))):
    n = len(d["offsets"]) - 1
    ctx = _lib.Context(0)
    ctx.upload_pool(d["frames"], d["offsets"])
    pairs = oracle.all_pairs(n, True).astype(np.int32)
    out = {}
    for ar in ("exact", "f16x2"):
        p = _lib.serra09_params(arith=ar)
        ctx.serra09_pairs(pairs[:256], p)
        t0 = time.time(); s = ctx.serra09_pairs(pairs, p); dt = time.time() - t0
        D = np.zeros((n, n), np.float32); D[pairs[:, 0], pairs[:, 1]] = s; D += D.T
        D /= np.sqrt(np.diff(d["offsets"]).astype(np.float64))[None, :]
        st = oracle.eval_statistics(D.astype(np.float32), cliques(d["labels"]), topsidx=(1, 10))
        out[ar] = (s, st, len(pairs) / dt)
    se, sf = out["exact"][0], out["f16x2"][0]
    diff = np.abs(se - sf)
    print("%s: %d pairs | exact %.0f k pairs/s, f16x2 %.0f k (x %.3f) | identical %.4f, |diff| <= 2: %.5f, max %.1f | MAP %.6f / %.6f  MR %.4f / %.4f  top1 %d / %d" % (
        name, len(pairs), out["exact"][2] / 1e3, out["f16x2"][2] / 1e3, out["f16x2"][2] / out["exact"][2], np.mean(diff == 0), np.mean(diff <= 2.0), diff.max(),
        out["exact"][1][3], out["f16x2"][1][3], out["exact"][1][0], out["f16x2"][1][0], out["exact"][1][4][0], out["f16x2"][1][4][0]))
    ctx.close()
