"""Ad-hoc timing of the similarity-network-fusion post-step (acx_snf_fuse_dists) at DA-TACOS size.
usage: python scripts/bench_snf.py [n] [m]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(0)
Ds = []
for _ in range(m):
    D = rng.random((n, n), dtype=np.float32).astype(np.float64)
    Ds.append(D + D.T)
ctx = _lib.Context(0)
for rep in range(2):
    t0 = time.time()
    _, F = ctx.snf_fuse_dists(Ds, K=20, niters=20, reg_diag=1.0)
    print("n=%d m=%d: %.2f s (affinity matrices, kNN lists, 20 sweeps, transfers)" % (n, m, time.time() - t0))
ctx.close()
