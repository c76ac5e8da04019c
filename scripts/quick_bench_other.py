"""Ad-hoc timing of the SiMPle and EarlyFusion paths at DA-TACOS-like shapes (development aid)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib, synth  # noqa: E402

ctx = _lib.Context(0)
ctx.profile_enable(True)
rng = np.random.default_rng(0)
# SiMPle: 12 x 150-250 pooled frames
n = 256
feats = [rng.random((int(rng.integers(150, 251)), 12)) for _ in range(n)]
offs = np.concatenate([[0], np.cumsum([len(f) for f in feats])]).astype(np.int64)
ctx.upload_pool_f64(np.concatenate(feats), offs)
i, j = np.nonzero(~np.eye(n, dtype=bool))
pairs = np.stack([i, j], 1).astype(np.int32)
ctx.simple_pairs(pairs[:1000])
t0 = time.time(); out = ctx.simple_pairs(pairs); dt = time.time() - t0
print("SiMPle: %d ordered pairs %.3f s  %.0f pairs/s" % (len(pairs), dt, len(pairs) / dt))
# EarlyFusion: 300-500 blocks
n = 24
tracks = synth.earlyfusion_set(n, seed=1, nb_range=(300, 500))
ctx.ef_upload_pool(tracks)
i, j = np.triu_indices(n, 1)
pairs = np.stack([i, j], 1).astype(np.int32)
ctx.earlyfusion_pairs(pairs[:8])
ctx.profile_reset()
t0 = time.time(); out = ctx.earlyfusion_pairs(pairs); dt = time.time() - t0
print("EarlyFusion: %d pairs %.3f s  %.0f pairs/s" % (len(pairs), dt, len(pairs) / dt))
for k, v in ctx.profile().items():
    if v["launches"]:
        print("   %-18s %9.3f ms  %3d launches" % (k, v["ms"], v["launches"]))
