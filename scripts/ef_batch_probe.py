import sys, time
sys.path.insert(0, ".")
import numpy as np
from acoss_amd import _lib, synth
ctx = _lib.Context(0)
tracks = synth.earlyfusion_set(300, seed=1, nb_range=(300, 500))
ctx.ef_upload_pool(tracks)
n = 300
for lim in (0, 24 << 30, 8 << 30):
    ctx.set_scratch_limit(lim)
    for rep in range(2):
        planes = [np.zeros((n, n), np.float32) for _ in range(4)]
        t0 = time.time()
        ctx.pair_grid(_lib.ALGO_EARLYFUSION, True, _lib.EfParams(0.1, 10), planes, mirror=True)
        dt = time.time() - t0
        print("scratch limit %5.1f GB rep %d: %.2f s, %.0f pairs/s" % (lim / 2**30, rep, dt, n * (n - 1) / 2 / dt))
