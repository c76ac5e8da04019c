"""Ad-hoc timing of the Serra09 chain (development aid; bench.py is the contract)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
d = synth.rand_set(n, T=T, seed=1234)
ctx = _lib.Context(0)
ctx.upload_pool(d["frames"], d["offsets"])
i, j = np.triu_indices(n, 1)
pairs = np.stack([i, j], 1).astype(np.int32)
ctx.serra09_pairs(pairs[:64])
ctx.profile_enable(True)
for rep in range(2):
    ctx.profile_reset()
    t0 = time.time()
    out = ctx.serra09_pairs(pairs)
    dt = time.time() - t0
    import zlib
    print("   scores crc32 %08x" % zlib.crc32(np.ascontiguousarray(out, dtype=np.float32).tobytes()))
    print("n=%d T=%d pairs=%d  %.3f s  %.1f pairs/s  (max score %.1f)" % (n, T, len(pairs), dt, len(pairs) / dt, out.max()))
    for k, v in ctx.profile().items():
        if v["launches"]:
            print("   %-18s %9.3f ms  %3d launches  %.2f ns/cell" % (k, v["ms"], v["launches"], 1e6 * v["ms"] / max(1, v["cells"])))
