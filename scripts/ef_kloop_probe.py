"""How fast is the k loop of the rectangle GEMM by itself?  All pairs of 48 tracks of 384 blocks (full tiles) with an mfcc
feature of K0 and of K0 + 3200 values: the difference of the GEMM times over the difference of the k chunks is the
time of one chunk without prologue / epilogue.  (python scripts/ef_kloop_probe.py)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(ctx, k0, n=48, nb=384, steps=3):
    import torch
    from acoss_amd import _lib
    rng = np.random.default_rng(1)
    tracks = [dict(mfccs=rng.standard_normal((nb, k0)).astype(np.float32), ssms=rng.random((nb, 32)).astype(np.float32),
                   chromas=rng.random((nb, 96)).astype(np.float32), chroma_med=rng.random(12)) for _ in range(n)]
    ctx.ef_upload_pool(tracks)
    lens = np.full(n, nb)
    plan = _lib.grid_plan(lens, _lib.ALGO_EARLYFUSION, True, world=1, tile=128, want_tiles=True)
    buf = torch.zeros(int(plan["floats_per_rank"][0]), dtype=torch.float32, device="cuda:0")
    ep = _lib.EfParams(0.1, 10)
    ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr())
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(steps):
        ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr())
    return ctx.profile()["ef_gemm_kernel"]["ms"] / steps


def main():
    from acoss_amd import _lib
    ctx = _lib.Context(0)
    a = run(ctx, 640)
    b = run(ctx, 640 + 3200)
    n, nb = 48, 384
    pairs = n * (n - 1) // 2
    flops = 2.0 * 3200 * nb * nb * pairs * 6                 # bf16 flops of the extra chunks (six products per cell)
    print(json.dumps({"gemm_ms_k640": round(a, 3), "gemm_ms_k3840": round(b, 3),
                      "bf16_tflops_of_the_extra_chunks": round(flops / ((b - a) * 1e-3) / 1e12, 1),
                      "f32eq_tflops_of_the_extra_chunks": round(flops / 6 / ((b - a) * 1e-3) / 1e12, 1)}))
    ctx.close()


if __name__ == "__main__":
    main()
