import sys
import numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib
rng = np.random.default_rng(1)
T = 520
for tag, quant in (("fp16-exact inputs (multiples of 1/64)", True), ("random f32 inputs", False)):
    q = rng.random((T, 12), dtype=np.float32); r = rng.random((T, 12), dtype=np.float32)
    if quant:
        q = np.round(q * 64) / 64; r = np.round(r * 64) / 64
    q = q.astype(np.float32); r = r.astype(np.float32)
    ctx = _lib.Context(0)
    ctx.upload_pool(np.concatenate([q, r]), np.array([0, T, 2 * T]))
    e = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False))["d2"]
    f = ctx.serra09_debug_pair(0, 1, _lib.serra09_params(oti=False, arith="f16x2"))["d2"]
    bad = np.abs(f - e) > 1e-3 * (1 + np.abs(e))
    print(tag, ": bad", int(bad.sum()), "of", bad.size, " row0 first:", np.round(e[0, :4], 3), np.round(f[0, :4], 3))
    ctx.close()
