"""Ad-hoc timing of the Serra09 chain on a covers80-shaped set (164 tracks, T ~ U{300..600})."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib, synth  # noqa: E402

sizes = [2] * 77 + [3, 3, 4]
t_range = (300, 600)
if len(sys.argv) > 2:                      # quick_bench_covers.py n_works t_lo t_hi
    sizes = [2] * int(sys.argv[1])
    t_range = (int(sys.argv[2]), int(sys.argv[3]))
d = synth.cover_set(clique_sizes=sizes, seed=4321, t_range=t_range)
n = len(d["offsets"]) - 1
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
    L = np.diff(d["offsets"]) - 9
    cells = float(np.sum(L[pairs[:, 0]] * L[pairs[:, 1]]))
    print("n=%d pairs=%d  %.3f s  %.1f pairs/s  %.1f Gcells/s" % (n, len(pairs), dt, len(pairs) / dt, cells / dt / 1e9))
    for k, v in ctx.profile().items():
        if v["launches"]:
            print("   %-18s %9.3f ms  %3d launches" % (k, v["ms"], v["launches"]))
