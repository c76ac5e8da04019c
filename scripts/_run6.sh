export PYTHONUNBUFFERED=1 TMPDIR=/tmp
python3 - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
from acoss_amd import _lib, synth
ctx = _lib.Context(0)
for works in (82, 200, 400):
    d = synth.cover_set(clique_sizes=[2] * works, seed=4321, t_range=(150, 650))
    n = len(d["offsets"]) - 1
    ctx.upload_pool(d["frames"], d["offsets"])
    i, j = np.triu_indices(n, 1)
    pairs = np.stack([i, j], 1).astype(np.int32)
    L = (np.diff(d["offsets"]) - 9).astype(np.float64)
    cells = float(np.sum(L[i] * L[j]))
    ctx.serra09_pairs(pairs[:64])
    ts = []
    for _ in range(5):
        t0 = time.time(); ctx.serra09_pairs(pairs); ts.append(time.time() - t0)
    print("covers 150-650 n=%d pairs=%d  best %.3f ms  %.0f k pairs/s  %.1f Gcells/s" % (n, len(pairs), min(ts) * 1e3, len(pairs) / min(ts) / 1e3, cells / min(ts) / 1e9), flush=True)
PY
