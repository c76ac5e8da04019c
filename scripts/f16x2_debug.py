import sys
import numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib, synth
rng = np.random.default_rng(1)
for T, shift in ((80, 0), (80, 5), (300, 0), (300, 7), (600, 3), (1200, 2), (2000, 0), (2000, 9)):
    q = rng.random((T, 12), dtype=np.float32); q /= q.max(axis=1, keepdims=True)
    r = np.roll(q, shift, axis=1) + 0.3 * rng.random((T, 12), dtype=np.float32); r /= r.max(axis=1, keepdims=True)
    ctx = _lib.Context(0)
    ctx.upload_pool(np.concatenate([q, r]), np.array([0, T, 2 * T]))
    e = ctx.serra09_debug_pair(0, 1, _lib.serra09_params())
    f = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(arith="f16x2"))
    rel = np.abs(f["d2"] - e["d2"]) / (np.abs(e["d2"]) + 1.0)
    bad = np.argwhere(rel > 1e-4)
    print("T=%d shift=%d oti %d/%d: max rel %.3g, cells off %d of %d; first bad %s; score %.1f / %.1f" % (T, shift, e["oti"], f["oti"], rel.max(), len(bad), rel.size, bad[:3].tolist(), e["score"], f["score"]))
    ctx.close()
