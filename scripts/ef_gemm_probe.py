"""EarlyFusion GEMM probe: kernel time of the three cross-similarity GEMMs for all pairs of n tracks of 300-500
blocks (python scripts/ef_gemm_probe.py [n] [steps]); prints one JSON line.  Used for A/B runs of build / env variants."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import torch
    from acoss_amd import _lib, synth
    ctx = _lib.Context(0)
    tracks = synth.earlyfusion_set(n, seed=1, nb_range=(300, 500))
    ctx.ef_upload_pool(tracks)
    if os.environ.get("ACX_PROBE_MODE"):
        ctx.set_ef_gemm(os.environ["ACX_PROBE_MODE"])
    nb = np.array([t["mfccs"].shape[0] for t in tracks])
    plan = _lib.grid_plan(nb, _lib.ALGO_EARLYFUSION, True, world=1, tile=128, want_tiles=True)
    buf = torch.zeros(int(plan["floats_per_rank"][0]), dtype=torch.float32, device="cuda:0")
    ep = _lib.EfParams(0.1, 10)
    ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr())
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr())
    dt = (time.perf_counter() - t0) / steps
    prof = ctx.profile()
    i, j = np.triu_indices(n, 1)
    flops = float(np.sum(2.0 * (650 + 1225 + 480) * nb[i] * nb[j]))
    kms = prof["ef_gemm_kernel"]["ms"] / steps
    print(json.dumps({"n": n, "pairs": len(i), "pairs_per_s": round(len(i) / dt, 1), "ms_per_step": round(1e3 * dt, 3),
                      "gemm_ms": round(kms, 3), "gemm_tflops_f32eq": round(flops / kms / 1e9, 1),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("ACX_")},
                      "kernels_ms": {k: round(v["ms"] / steps, 3) for k, v in prof.items() if v["launches"]},
                      "checksum": float(buf.double().sum().item())}))
    ctx.close()


if __name__ == "__main__":
    main()
