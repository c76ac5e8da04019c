#!/usr/bin/env python3
"""
bench_other.py -- the other two algorithms of the path at DA-TACOS-like shapes (BASELINE.json
configs[3] and [4] per-track shapes; one GPU).  NOT the driver's contract benchmark (that is
bench.py, Serra09): a companion that prints one JSON line per algorithm with the same fields --
throughput in track-pairs/s with inputs resident in HBM, the dominant kernel against its
roofline (SURVEY 8d flop models), and the CPU oracle timed on a bounded sample and compared.

    python bench_other.py [--steps K] [--warmup W]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TF = 157.3       # MI355X_MICROARCH.md: f32-input MFMA = f32 vector rate
F64_VALU_PEAK_TF = 78.6        # f64 vector FMA peak (half the f32 rate)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-pairs", type=int, default=4)
    args = ap.parse_args()
    from acoss_amd import _lib, synth
    import oracle                      # only the cpu_baseline legs below use it (the checker, never the measured path)
    ctx = _lib.Context(0)
    rng = np.random.default_rng(0)

    # ---------------- SiMPle: 12 x 150-250 pooled frames, ordered pairs (simple_silva.py:120-126)
    n = 512
    from acoss_amd.algorithms.simple_silva import Simple
    feats = [Simple.smooth(None, rng.random((12, int(rng.integers(150, 251))))) for _ in range(n)]   # product host code
    tm = [np.ascontiguousarray(f.T) for f in feats]                 # time-major for the pool
    offs = np.concatenate([[0], np.cumsum([len(f) for f in tm])]).astype(np.int64)
    ctx.upload_pool_f64(np.concatenate(tm), offs)
    i, j = np.nonzero(~np.eye(n, dtype=bool))
    pairs = np.stack([i, j], 1).astype(np.int32)
    for _ in range(args.warmup):
        ctx.simple_pairs(pairs)
    ctx.profile_enable(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = ctx.simple_pairs(pairs)
    dt = (time.perf_counter() - t0) / args.steps
    prof = ctx.profile()
    na = np.array([len(f) for f in tm])
    cells = float(np.sum((na[pairs[:, 0]] - 9.0) * (na[pairs[:, 1]] - 9.0)))
    # algorithmic f64 work per profile cell = the reference's STOMP update (simple_silva.py:107-110): two 12-term
    # products (the frame entering and the frame leaving the window) + the update, the distance and the running
    # minimum = 54 flop.  (The kernel EXECUTES 35.6 of them: the leaving product is handed over by the lane 10 below
    # instead of being recomputed, on 64 lanes of which 54 are rows.)  SURVEY 8d's model (a 120-term product per
    # cell, 240 flop) is reported beside it
    flops = 54.0 * cells
    executed_flops = 30.0 * 64.0 / 54.0 * cells
    model_flops = 240.0 * cells
    kms = prof["simple_kernel"]["ms"] / max(1, prof["simple_kernel"]["launches"])
    ncpu = min(64, len(pairs))
    tc = time.perf_counter()
    ref = np.array([oracle.simple_pair(feats[a], feats[b]) for a, b in pairs[:ncpu]])
    tcpu = time.perf_counter() - tc
    err = float(np.max(np.abs(ref - out[:ncpu])))
    assert err < 1e-9 * max(1.0, float(np.max(np.abs(ref)))), err
    print(json.dumps({
        "metric": "ordered track-pairs/sec, SiMPle (matrix profile median) on 12 x 150-250 pooled frames",
        "value": round(len(pairs) / dt, 1), "unit": "track-pairs/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt, 3), "higher_is_better": True, "dtype": "f64",
        "data": "synthetic", "config": {"workload": "%d tracks, %d ordered pairs per step" % (n, len(pairs))},
        "roofline": {"bound": "scalar-cache latency (valu-f64 peak as the yardstick)", "kernel": "simple_kernel", "achieved": round(flops / (kms * 1e-3) / 1e12, 2),
                     "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": round(flops / (kms * 1e-3) / 1e12 / F64_VALU_PEAK_TF, 4),
                     "traffic": None, "avg_launch_ms": round(kms, 3),
                     "model_tflops_survey_8d": round(model_flops / (kms * 1e-3) / 1e12, 2),
                     "executed_tflops": round(executed_flops / (kms * 1e-3) / 1e12, 2),
                     "note": "achieved = the reference's STOMP update, 54 f64 flop per profile cell (two 12-term products + update); "
                             "the kernel executes 35.6 of them (executed_tflops: the leaving product is handed over by the lane 10 "
                             "below); SURVEY 8d's 120-term model would read model_tflops_survey_8d; the wave waits for the "
                             "scalar-cache misses of the streamed frames (s_waitcnt lgkmcnt(0) per step), not for the f64 pipe; "
                             "HBM negligible"},
        "cpu_baseline": {"value": round(ncpu / tcpu, 2), "unit": "track-pairs/s", "cores": 1, "kind": "port",
                         "sample": "first %d pairs, numpy oracle; max |diff| vs GPU %.2e" % (ncpu, err)}}))

    # ---------------- EarlyFusion: 300-500 blocks per track (earlyfusion_traile.py:157-198)
    n = 48
    tracks = synth.earlyfusion_set(n, seed=1, nb_range=(300, 500))
    ctx.ef_upload_pool(tracks)
    i, j = np.triu_indices(n, 1)
    pairs = np.stack([i, j], 1).astype(np.int32)
    for _ in range(args.warmup):
        ctx.earlyfusion_pairs(pairs)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = ctx.earlyfusion_pairs(pairs)
    dt = (time.perf_counter() - t0) / args.steps
    prof = ctx.profile()
    nb = np.array([t["mfccs"].shape[0] for t in tracks])
    flops = float(np.sum(2.0 * (650 + 1225 + 480) * nb[pairs[:, 0]] * nb[pairs[:, 1]]))
    g = prof["ef_gemm_kernel"]
    kms = g["ms"] / max(1, g["launches"])
    ncpu = min(args.cpu_pairs, len(pairs))
    tc = time.perf_counter()
    ref = []
    for a, b in pairs[:ncpu]:
        sc = oracle.earlyfusion_pair(tracks[a], tracks[b])[0]
        ref.append([sc["mfccs"], sc["ssms"], sc["chromas"], sc["early"]])
    ref = np.array(ref)
    tcpu = time.perf_counter() - tc
    diff = float(np.max(np.abs(ref - out[:ncpu])))
    print(json.dumps({
        "metric": "track-pairs/sec, EarlyFusion per-pair chain (3 CSMs, 4 x binarise + Smith-Waterman, kernel fusion) at 300-500 blocks",
        "value": round(len(pairs) / dt, 1), "unit": "track-pairs/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt, 3), "higher_is_better": True, "dtype": "f32",
        "data": "synthetic", "config": {"workload": "%d tracks, %d pairs per step" % (n, len(pairs))},
        "roofline": {"bound": "mfma", "kernel": "ef_gemm_kernel", "achieved": round(flops / (kms * 1e-3) / 1e12, 2),
                     "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(flops / (kms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF, 4),
                     "traffic": None, "avg_launch_ms": round(kms, 3),
                     "kernels_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if v["launches"]}},
        "cpu_baseline": {"value": round(ncpu / tcpu, 3), "unit": "track-pairs/s", "cores": 1, "kind": "port",
                         "sample": "first %d pairs, numpy + C oracle; max |score diff| vs GPU %.3g" % (ncpu, diff)}}))
    ctx.close()


if __name__ == "__main__":
    main()
