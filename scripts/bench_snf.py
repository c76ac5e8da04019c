"""Ad-hoc timing of the device SNF loop (acx_snf_fuse).  usage: bench_snf.py [n] [m]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib  # noqa: E402
from acoss_amd.algorithms import similarity_fusion as sf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(0)
ctx = _lib.Context(0)
Scores = [rng.random((n, n)) for _ in range(m)]
t0 = time.time()
Ws = [sf.getW(D, 20) for D in Scores]
t1 = time.time()
lists = [sf._knn_lists(W, 20) for W in Ws]
t2 = time.time()
out = ctx.snf_fuse(Ws, [l[0] for l in lists], [l[1] for l in lists], 20, 1.0)
t3 = time.time()
print("n=%d m=%d: getW %.2f s, neighbour lists %.2f s (host), device loop (20 sweeps, incl. transfers) %.2f s" % (n, m, t1 - t0, t2 - t1, t3 - t2))
print("checksum %.6f" % float(out.sum()))
