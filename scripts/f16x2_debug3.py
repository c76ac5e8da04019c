import sys
import numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib
rng = np.random.default_rng(1)
for T in (520, 1100):
    q = rng.random((T, 12), dtype=np.float32); q /= q.max(axis=1, keepdims=True)
    r = rng.random((T, 12), dtype=np.float32); r /= r.max(axis=1, keepdims=True)
    ctx = _lib.Context(0)
    ctx.upload_pool(np.concatenate([q, r]), np.array([0, T, 2 * T]))
    e = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False))
    f = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False, arith="f16x2"))
    bad = np.abs(f["d2"] - e["d2"]) > 1e-3 * (1 + np.abs(e["d2"]))
    for row in (0, 3, 7, 8):
        b = bad[row]
        # runs of good columns
        good = np.nonzero(~b)[0]
        runs = []
        if len(good):
            s = good[0]; p = s
            for g in good[1:]:
                if g != p + 1: runs.append((int(s), int(p))); s = g
                p = g
            runs.append((int(s), int(p)))
        print("T=%d row %d: good column runs %s" % (T, row, runs[:12]))
    ctx.close()
