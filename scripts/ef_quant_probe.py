"""Development aid: how much of ef_gemm's time is tile quantisation?  The per-pair GEMM tiles a (M x N) matrix in
128 x 128 workgroup tiles; M = N = 384 fills 9 tiles, 400 needs 16 (7 of them 16 rows / columns wide).
usage: python scripts/ef_quant_probe.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from acoss_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
rng = np.random.default_rng(0)
for nb in (256, 384, 400, 448, 512):
    n = 32
    tracks = []
    for _ in range(n):
        mf = rng.standard_normal((nb, 650)).astype(np.float32)
        mf /= np.linalg.norm(mf, axis=1, keepdims=True)
        tracks.append(dict(mfccs=mf, ssms=(2 * rng.random((nb, 1225))).astype(np.float32), chromas=rng.random((nb, 480)).astype(np.float32),
                           chroma_med=rng.random(12)))
    ctx.ef_upload_pool(tracks)
    i, j = np.triu_indices(n, 1)
    pairs = np.stack([i, j], 1).astype(np.int32)
    ctx.earlyfusion_pairs(pairs)
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.earlyfusion_pairs(pairs)
    dt = (time.perf_counter() - t0) / 3
    prof = ctx.profile()
    g = prof["ef_gemm_kernel"]["ms"] / 3
    flops = 2.0 * 2355 * nb * nb * len(pairs)
    print("nb %4d: %7.1f pairs/s  gemm %.3f ms = %.1f TFLOP/s f32-equivalent; rowstat %.3f fuse %.3f sw %.3f ms" % (
        nb, len(pairs) / dt, g, flops / g / 1e9, prof["ef_rowstat_kernel"]["ms"] / 3, prof["ef_fuse_kernel"]["ms"] / 3, prof["sw_kernel"]["ms"] / 3))
ctx.close()
