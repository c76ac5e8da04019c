import sys
import numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib
rng = np.random.default_rng(1)
for T in (500, 520, 600):
    q = rng.random((T, 12), dtype=np.float32); q /= q.max(axis=1, keepdims=True)
    r = rng.random((T, 12), dtype=np.float32); r /= r.max(axis=1, keepdims=True)
    ctx = _lib.Context(0)
    ctx.upload_pool(np.concatenate([q, r]), np.array([0, T, 2 * T]))
    e = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False))
    f = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False, arith="f16x2"))
    bad = np.abs(f["d2"] - e["d2"]) > 1e-3 * (1 + np.abs(e["d2"]))
    print("T=%d: bad cells %d of %d; bad per row (first 10 rows) %s" % (T, bad.sum(), bad.size, bad[:10].sum(axis=1).tolist()))
    cols = np.nonzero(bad.any(axis=0))[0]
    if len(cols):
        print("   bad columns: %d..%d, count %d; row 0 bad cols %s" % (cols.min(), cols.max(), len(cols), np.nonzero(bad[0])[0][:20].tolist()))
        print("   exact", e["d2"][0, :6], "\n   f16  ", f["d2"][0, :6])
        print("   finite fraction of f16:", np.isfinite(f["d2"]).mean(), "eps_q equal:", np.array_equal(e["eps_q"], f["eps_q"]))
    ctx.close()
