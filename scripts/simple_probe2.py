"""Development aid: does simple_kernel wait for scalar-cache misses of the streamed track?  Same work, but (a) every pair
streams ONE second track (its frames stay in the scalar cache), (b) second tracks as in the grid (sorted), (c) random."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
rng = np.random.default_rng(0)
n = 2048
T = 200
F = rng.random((n * T, 12))
F /= np.linalg.norm(F, axis=1, keepdims=True)
offs = np.arange(n + 1, dtype=np.int64) * T
ctx.upload_pool_f64(F, offs)
K = 1 << 20
first = rng.integers(1, n, K).astype(np.int32)
cases = {"one second track": np.stack([first, np.zeros(K, np.int32)], 1),
         "sorted second tracks": np.stack([first, np.sort(rng.integers(0, n, K)).astype(np.int32)], 1),
         "random second tracks": np.stack([first, rng.integers(0, n, K).astype(np.int32)], 1)}
for name, pairs in cases.items():
    pairs = np.ascontiguousarray(pairs[pairs[:, 0] != pairs[:, 1]])
    ctx.simple_pairs(pairs[:1000])
    ctx.profile_enable(True)
    ctx.profile_reset()
    ctx.simple_pairs(pairs)
    ms = ctx.profile()["simple_kernel"]["ms"]
    print("%-22s %8.2f ms kernel, %.1f M pairs/s" % (name, ms, len(pairs) / ms / 1e3))
ctx.close()
