"""EarlyFusion grid tile of 128 tracks at different scratch limits = pairs per batch: do batches whose matrices fit the 256 MB
Infinity Cache between the kernels of the chain run faster?  (No: DESIGN.md 5c.)  python scripts/ef_mall_probe.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acoss_amd import _lib, synth
n = 128
ctx = _lib.Context(0)
tracks = synth.earlyfusion_set(n, seed=1, nb_range=(300, 500))
ctx.ef_upload_pool(tracks)
nb = np.array([t["mfccs"].shape[0] for t in tracks])
plan = _lib.grid_plan(nb, _lib.ALGO_EARLYFUSION, True, world=1, tile=128, want_tiles=True)
buf = torch.zeros(int(plan["floats_per_rank"][0]), dtype=torch.float32, device="cuda:0")
ep = _lib.EfParams(0.1, 10)
for lim_mb in (0, 160, 240, 400, 800, 2000, 8000):
    ctx.set_scratch_limit(lim_mb << 20)
    ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr())
    ctx.profile_enable(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr())
    dt = (time.perf_counter() - t0) / 3
    prof = ctx.profile()
    print(json.dumps({"scratch_limit_MB": lim_mb, "pairs_per_s": round(8128 / dt), "ms_per_step": round(1e3 * dt, 2),
                      "kernels_ms": {k: round(v["ms"] / 3, 2) for k, v in prof.items() if v["launches"]}, "checksum": float(buf.double().sum().item())}))
