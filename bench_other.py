#!/usr/bin/env python3
"""
bench_other.py -- the companion legs of bench.py (one GPU): Serra09 on the track lengths real collections have
(BASELINE.json configs[1] shape), the opt-in f16x2 Gram beside the exact one on the headline's shape, and the other two
algorithms at DA-TACOS-like per-track shapes (configs[3] and [4]).  bench.py (the driver's contract benchmark: Serra09 on
configs[2]) calls the legs below after its own timed region and carries their results in the `other` object of its JSON
line; run on its own this file prints one JSON line per leg with the same fields:

    python bench_other.py [--steps K] [--warmup W]

Per leg: throughput in track-pairs/s with inputs resident in HBM, the dominant kernel against its roofline (HIP events on
the library's stream), and the CPU oracle timed on a bounded sample (one thread) and compared with the GPU's scores.  No
torch in the process: device buffers come from libacx (acx_dev_alloc).
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
BF16_MFMA_PEAK_TF = 2500.0     # dense bf16 matrix peak (no sparsity)


def _ref_ratio(key):
    """The oracle's per-pair speed relative to the REFERENCE's own numpy code, measured once in the authoring container
    (scripts/ref_vs_oracle_timing.py -> profiles/r05_ref_vs_oracle.json: the reference never travels to the GPU box)."""
    try:
        r = json.load(open(os.path.join(ROOT, "profiles", "r05_ref_vs_oracle.json")))[key]["oracle_speed_over_reference"]
        return ("; this port runs at %.2f x the per-pair, one-thread speed of the reference's own numpy code (profiles/r05_ref_vs_oracle.json, "
                "taken in the authoring container)" % r)
    except Exception:                               # noqa: BLE001
        return ""


def other_source_sha16():
    """Hash of the kernel sources the companion legs run (scripts/make_traffic_other.py stamps its record with it)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("serra09_kernels.hpp", "serra09_band2_kernels.hpp", "acx_band.hip", "ef_kernels.hpp", "ef_gemm_persist_kernels.hpp", "ef_gemm_dma_kernels.hpp", "ef_rowstat2_kernels.hpp", "simple_kernels.hpp", "Makefile"):
        with open(os.path.join(ROOT, "acoss_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _traffic(leg):
    """(HBM bytes per launch of the leg's dominant kernel family, where it comes from) from profiles/pmc_traffic_other.json --
    counters of a SEPARATE rocprofv3 --pmc run, reported only for the kernel sources of this build; else (None, why)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_other.json")))
    except Exception:                               # noqa: BLE001
        return None, "no profiles/pmc_traffic_other.json"
    if tj.get("kernel_source_sha16") != other_source_sha16():
        return None, "profiles/pmc_traffic_other.json was taken on other kernel sources: not reported"
    if leg not in tj:
        return None, "no record for this leg"
    return tj[leg]["hbm_bytes_per_launch"], "profiles/pmc_traffic_other.json (%s; mean over the launches of the kernel family)" % tj.get("source")


def _one_thread():
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1)
    except ImportError:
        import contextlib
        return contextlib.nullcontext()


def _tile_pairs(t):
    """(row, col) track pairs of one grid tile, in the row-major order of its cells in the device buffer."""
    return [(t.row0 + a, t.col0 + b) for a in range(t.rows) for b in range(t.cols)
            if (t.row0 + a < t.col0 + b if t.diagonal else t.row0 + a != t.col0 + b)]


def serra09_covers_leg(ctx, steps=5, warmup=1, cpu_pairs=192):
    """Serra09 on the lengths real collections have (covers80 tracks are ~150-650 pooled
    frames; rqa_serra09.py:44-69): `steps + warmup` covers80-shaped sets (164 tracks / 80 works each, cover structure,
    T ~ U{150..650}, one seed per step) in one pool, all 13 366 pairs of a different set per step through
    acx_serra09_pairs (the pair-list entry point similarity() uses); and once, beside it, a DA-TACOS-like length mix
    (no length statistics of DA-TACOS are available offline: log-normal around 450 frames, sigma 0.35, clipped to
    120 .. 1600 -- an assumption, stated).  Cells per second against the T = 2000 rate is the figure to watch: the
    selection's per-row chain does not shrink with the row."""
    import oracle
    from acoss_amd import _lib, synth
    sets = [synth.covers80_shaped(seed=100 + s, t_range=(150, 650)) for s in range(steps + warmup)]
    frames = np.concatenate([d["frames"] for d in sets])
    lens = np.concatenate([np.diff(d["offsets"]) for d in sets])
    # the DA-TACOS-like mix rides in the same pool, after the covers80-shaped sets
    rng = np.random.default_rng(77)
    mix_T = np.clip(np.round(np.exp(rng.normal(np.log(450.0), 0.35, 328))), 120, 1600).astype(np.int64)
    mix = synth.cover_set(clique_sizes=[2] * 164, seed=78, t_range=(150, 650))       # structure donor: re-cut to mix_T
    mix_frames = []
    mfr, moff = mix["frames"], mix["offsets"]
    for k, T in enumerate(mix_T):
        src_ = mfr[moff[k]:moff[k + 1]]
        mix_frames.append(src_[np.arange(T) % len(src_)])
    frames = np.concatenate([frames] + mix_frames)
    lens = np.concatenate([lens, mix_T])
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ctx.upload_pool(frames, offs)
    params = _lib.serra09_params()
    n = 164
    iu, ju = np.triu_indices(n, 1)
    base = np.stack([iu, ju], 1).astype(np.int32)
    L = lens - 9

    def cells_of(p):
        return float(np.sum(L[p[:, 0]].astype(np.float64) * L[p[:, 1]]))

    for s in range(warmup):
        ctx.serra09_pairs(base + s * n, params)
    # The timed steps run WITHOUT the library's per-kernel event clocks (two hipEventRecord around every launch: a call of
    # 13 366 short pairs is ~9 ms of kernels in ~10 launches, the clocks cost it 3-8 %); the kernel times of the roofline
    # come from a second, untimed pass over the same steps with the clocks on.
    ctx.profile_enable(False)
    dt, cells, got0 = 0.0, 0.0, None
    for s in range(warmup, warmup + steps):
        p = np.ascontiguousarray(base + s * n)
        t0 = time.perf_counter()
        got = ctx.serra09_pairs(p, params)
        dt += time.perf_counter() - t0
        cells += cells_of(p)
        if got0 is None:
            got0, p0 = got, p
    ctx.profile_enable(True)
    ctx.profile_reset()
    for s in range(warmup, warmup + steps):
        ctx.serra09_pairs(np.ascontiguousarray(base + s * n), params)
    prof = ctx.profile()
    ctx.profile_enable(False)
    npairs = steps * len(base)
    # the length mix: all pairs of its 328 tracks, twice (first call untimed)
    m0 = (steps + warmup) * n
    im, jm = np.triu_indices(len(mix_T), 1)
    pm = np.ascontiguousarray(np.stack([im + m0, jm + m0], 1).astype(np.int32))
    ctx.serra09_pairs(pm[:4096], params)
    t0 = time.perf_counter()
    gm = ctx.serra09_pairs(pm, params)
    dtm = time.perf_counter() - t0
    # CPU oracle, one core, bounded sample of the first timed set; bit-for-bit check of both workloads
    with _one_thread():
        tc = time.perf_counter()
        ref = oracle.serra09_pairs(frames, offs, p0[:cpu_pairs])
        tcpu = time.perf_counter() - tc
        refm = oracle.serra09_pairs(frames, offs, pm[:32])
    if not (np.array_equal(ref, got0[:cpu_pairs]) and np.array_equal(refm, gm[:32])):
        raise AssertionError("serra09_covers: GPU scores differ from the CPU oracle")
    kname, kst = max(prof.items(), key=lambda kv: kv[1]["ms"])
    kbytes = 4.0 * kst["cells"]                       # the byte model of bench.py: 4 B per cell and band_kernel launch
    return {
        "metric": "track-pairs/sec, Serra09 Qmax on covers80-shaped lengths (164 tracks of 150-650 pooled frames, all 13 366 pairs)",
        "value": round(npairs / dt, 1), "unit": "track-pairs/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(1e3 * dt / steps, 3), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1] shape: %d covers80-shaped sets (seeds 100..%d) in one pool, a different set per step, "
                               "acx_serra09_pairs on the pair list (scores to the host)" % (steps + warmup, 100 + steps + warmup - 1)},
        "gcells_per_s": round(cells / dt / 1e9, 1),
        "length_mix": {"workload": "328 tracks, lengths log-normal around 450 frames (sigma 0.35, clipped 120..1600: assumed, DA-TACOS "
                                   "length statistics are not available offline), all %d pairs in one call" % len(pm),
                       "value": round(len(pm) / dtm, 1), "gcells_per_s": round(cells_of(pm) / dtm / 1e9, 1)},
        "roofline": {"bound": "valu (selection: per-row latency chain) + f32 mfma", "kernel": kname,
                     "achieved": round(kbytes / (kst["ms"] * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(kbytes / (kst["ms"] * 1e-3) / 1e9 / 8000.0, 4), "traffic": _traffic("serra09_covers")[0],
                     "traffic_source": _traffic("serra09_covers")[1],
                     "model": "same 4 B per cell and launch byte model as the headline's band_kernel (a throughput proxy)",
                     "kernels_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in prof.items() if v["launches"]}},
        "cpu_baseline": {"value": round(cpu_pairs / tcpu, 2), "unit": "track-pairs/s", "cores": 1, "kind": "port",
                         "sample": "first %d pairs of the first timed set, C oracle (gcc -O3), one thread, %.1f s; scores "
                                   "bit-identical to the GPU's (also 32 pairs of the length mix)" % (cpu_pairs, tcpu)}}


def serra09_f16x2_leg(ctx, n=96, T=2000, reps=3):
    """The opt-in f16x2 Gram of the band kernel (include/acx.h ACX_ARITH_F16X2) beside the exact one on the headline's shape:
    all pairs of `n` i.i.d. tracks of `T` frames, both arithmetics, the same pair list -- pairs/s of each, how many scores
    agree, and on two pairs the fraction of recurrence-plot cells that flip (acx_serra09_debug_pair in both modes).  NOT part of
    the headline: the default and the timed region of bench.py stay on the exact arithmetic."""
    from acoss_amd import _lib, synth
    d = synth.rand_set(n, T=T, seed=4242)
    ctx.upload_pool(d["frames"], d["offsets"])
    i, j = np.triu_indices(n, 1)
    pairs = np.ascontiguousarray(np.stack([i, j], 1).astype(np.int32))
    res = {}
    for ar in ("exact", "f16x2"):
        p = _lib.serra09_params(arith=ar)
        ctx.serra09_pairs(pairs[:512], p)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            sc = ctx.serra09_pairs(pairs, p)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[ar] = (sc, len(pairs) / best)
    diff = np.abs(res["exact"][0] - res["f16x2"][0])
    flips = cells = 0
    for (a, b) in ((0, 1), (2, 3)):
        e = ctx.serra09_debug_pair(a, b, _lib.serra09_params())
        f = ctx.serra09_debug_pair(a, b, _lib.serra09_params(arith="f16x2"))
        Re = (e["d2"] <= e["thr_q"][:, None]) & (e["d2"] <= e["thr_r"][None, :])
        Rf = (f["d2"] <= f["thr_q"][:, None]) & (f["d2"] <= f["thr_r"][None, :])
        flips += int(np.sum(Re != Rf))
        cells += Re.size
    return {
        "metric": "track-pairs/sec, Serra09 Qmax with the opt-in f16x2 Gram (T = %d), beside the exact arithmetic on the same pairs" % T,
        "value": round(res["f16x2"][1], 1), "unit": "track-pairs/s", "n_gpus": 1, "steps": reps, "warmup": 1, "higher_is_better": True,
        "dtype": "f16 x 2 terms -> f32", "data": "synthetic",
        "config": {"workload": "%d i.i.d. tracks x %d frames (seed 4242), all %d pairs through acx_serra09_pairs, best of %d" % (n, T, len(pairs), reps)},
        "exact_value": round(res["exact"][1], 1), "speedup_vs_exact": round(res["f16x2"][1] / res["exact"][1], 3),
        "scores_identical_fraction": round(float(np.mean(diff == 0)), 4), "scores_within_2_fraction": round(float(np.mean(diff <= 2.0)), 5),
        "max_score_diff": float(diff.max()), "flipped_cell_fraction": flips / float(cells), "flipped_cells": [flips, cells],
        "note": "as accurate against f64 as the exact chain (scripts/f16x2_accuracy.py: rms 4.1e-6 on d^2 in both), not the same bits: "
                "a few cells per 10 000 change side of a threshold"}


def chenfusion_leg(ctx, steps=3, warmup=1, cpu_pairs=96):
    """LateFusionChen's pair work (latefusion_chen.py:58-73: Qmax AND Dmax of the same cross recurrence plot) on
    covers80-shaped lengths: `steps + warmup` sets of 164 tracks, all 13 366 pairs of a different set per step through
    acx_chenfusion_pairs; one-core CPU figure and bit-for-bit check on a bounded sample (the C oracle, both alignments)."""
    import oracle
    from acoss_amd import _lib, synth
    sets = [synth.covers80_shaped(seed=300 + s, t_range=(150, 650)) for s in range(steps + warmup)]
    frames = np.concatenate([d["frames"] for d in sets])
    lens = np.concatenate([np.diff(d["offsets"]) for d in sets])
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ctx.upload_pool(frames, offs)
    params = _lib.serra09_params()
    n = 164
    iu, ju = np.triu_indices(n, 1)
    base = np.stack([iu, ju], 1).astype(np.int32)
    L = lens - 9
    for s in range(warmup):
        ctx.chenfusion_pairs(base + s * n, params)
    ctx.profile_enable(False)                    # (timed without the event clocks, kernel times from a second pass: serra09_covers_leg)
    dt, cells, got0, p0 = 0.0, 0.0, None, None
    for s in range(warmup, warmup + steps):
        p = np.ascontiguousarray(base + s * n)
        t0 = time.perf_counter()
        got = ctx.chenfusion_pairs(p, params)
        dt += time.perf_counter() - t0
        cells += float(np.sum(L[p[:, 0]].astype(np.float64) * L[p[:, 1]]))
        if got0 is None:
            got0, p0 = got, p
    ctx.profile_enable(True)
    ctx.profile_reset()
    for s in range(warmup, warmup + steps):
        ctx.chenfusion_pairs(np.ascontiguousarray(base + s * n), params)
    prof = ctx.profile()
    ctx.profile_enable(False)
    with _one_thread():
        tc = time.perf_counter()
        rq = oracle.serra09_pairs(frames, offs, p0[:cpu_pairs], oracle.serra09_params())
        rd = oracle.serra09_pairs(frames, offs, p0[:cpu_pairs], oracle.serra09_params(dmax=1))
        tcpu = time.perf_counter() - tc
    if not (np.array_equal(rq, got0[:cpu_pairs, 0]) and np.array_equal(rd, got0[:cpu_pairs, 1])):
        raise AssertionError("chenfusion: GPU scores differ from the CPU oracle")
    npairs = steps * len(base)
    return {
        "metric": "track-pairs/sec, LateFusionChen's Qmax + Dmax on covers80-shaped lengths (164 tracks of 150-650 pooled frames, all 13 366 pairs)",
        "value": round(npairs / dt, 1), "unit": "track-pairs/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(1e3 * dt / steps, 3), "higher_is_better": True, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1] shape, %d sets (seeds 300..%d), a different set per step, acx_chenfusion_pairs (both scores to the host)"
                               % (steps + warmup, 300 + steps + warmup - 1)},
        "gcells_per_s": round(cells / dt / 1e9, 1),
        "kernels_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in prof.items() if v["launches"]},
        "cpu_baseline": {"value": round(cpu_pairs / tcpu, 2), "unit": "track-pairs/s", "cores": 1, "kind": "port",
                         "sample": "first %d pairs of the first timed set, C oracle (gcc -O3), one thread, Qmax and Dmax one after the "
                                   "other, %.1f s; both scores bit-identical to the GPU's" % (cpu_pairs, tcpu)}}


def simple_leg(ctx, steps=5, warmup=1, n=1536, tiles_per_step=16, cpu_pairs=64):
    """SiMPle (simple_silva.py:120-126): ordered pairs of tracks of 12 x 150-250 pooled frames through the pair grid
    (acx_grid_run: pairs enumerated on the device, f64 kernel, f32 scatter into a device buffer).  The pool holds
    `n` tracks; every step runs the next `tiles_per_step` 128 x 128 tiles of the grid (16 tiles = 262 144 ordered
    pairs, other tracks every step)."""
    import oracle                      # the checker / timed CPU baseline only
    from acoss_amd import _lib
    from acoss_amd.algorithms.simple_silva import Simple
    rng = np.random.default_rng(0)
    feats = [Simple.smooth(None, rng.random((12, int(rng.integers(150, 251))))) for _ in range(n)]   # product host code
    tm = [np.ascontiguousarray(f.T) for f in feats]                 # time-major for the pool
    na = np.array([len(f) for f in tm])
    offs = np.concatenate([[0], np.cumsum(na)]).astype(np.int64)
    ctx.upload_pool_f64(np.concatenate(tm), offs)
    plan = _lib.grid_plan(na, _lib.ALGO_SIMPLE, False, world=1, tile=128, want_tiles=True)
    tiles = list(plan["tiles"])
    assert len(tiles) >= (steps + warmup) * tiles_per_step
    buf = ctx.dev_alloc(4 * int(plan["floats_per_rank"][0]))          # (no torch in this process: libacx's own device buffer)
    sp = _lib.SimpleParams(10, 1)
    for s in range(warmup):
        ctx.grid_run(plan["spec"], sp, 0, buf.data_ptr(), first=s * tiles_per_step, count=tiles_per_step)
    ctx.profile_enable(False)                    # (timed without the event clocks, kernel times from a second pass: serra09_covers_leg)
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        ctx.grid_run(plan["spec"], sp, 0, buf.data_ptr(), first=s * tiles_per_step, count=tiles_per_step)   # returns after the stream has drained
    dt = (time.perf_counter() - t0) / steps
    ctx.profile_enable(True)
    ctx.profile_reset()
    for s in range(warmup, warmup + steps):
        ctx.grid_run(plan["spec"], sp, 0, buf.data_ptr(), first=s * tiles_per_step, count=tiles_per_step)
    prof = ctx.profile()
    ctx.profile_enable(False)
    host = buf.read(np.float32)
    buf.free()
    timed = tiles[warmup * tiles_per_step:(warmup + steps) * tiles_per_step]
    w = na - 9.0
    npairs, cells = 0, 0.0
    for t in timed:
        r, c = w[t.row0:t.row0 + t.rows], w[t.col0:t.col0 + t.cols]
        npairs += t.rows * t.cols - (t.rows if t.row0 == t.col0 else 0)
        cells += float(r.sum() * c.sum()) - (float(np.sum(r * c)) if t.row0 == t.col0 else 0.0)
    npairs /= float(steps)
    cells /= float(steps)
    # f64 work per profile cell.  The reference's STOMP update (simple_silva.py:107-110) is two 12-term products
    # (the frame entering and the frame leaving the window) + update + distance + running minimum = 54 flop; the
    # kernel EXECUTES one product (24 flop) + 6 per cell on 64 lanes of which 54 are rows = 35.6: the leaving
    # product is handed over by the lane SSLEN below.  SURVEY 8d's model (a 120-term product per cell, 240 flop)
    # counts work nobody does.  `achieved` / `frac` are the executed figure; the other two models have their own
    # frac_* fields.
    executed = 30.0 * 64.0 / 54.0 * cells
    kst = prof["simple_kernel"]
    kms = kst["ms"] / steps                                                    # kernel time per step
    ks = kms * 1e-3
    # CPU baseline + check: the first pairs of the first timed tile (row-major cells of the device buffer)
    t = timed[0]
    cp = [(t.row0 + a, t.col0 + b) for a in range(t.rows) for b in range(t.cols) if t.row0 + a != t.col0 + b][:cpu_pairs]
    with _one_thread():
        tc = time.perf_counter()
        ref = np.array([oracle.simple_pair(feats[a], feats[b]) for a, b in cp])
        tcpu = time.perf_counter() - tc
    got = np.array([host[t.offset + (a - t.row0) * t.cols + (b - t.col0)] for a, b in cp], np.float64)
    err = float(np.max(np.abs(got - ref) / np.abs(ref)))
    assert err < 2e-7, err                                                         # f32 store of the f64 result
    return {
        "metric": "ordered track-pairs/sec, SiMPle (matrix profile median) on 12 x 150-250 pooled frames",
        "value": round(npairs / dt, 1), "unit": "track-pairs/s", "n_gpus": 1, "steps": steps,
        "warmup": warmup, "ms_per_step": round(1e3 * dt, 3), "higher_is_better": True, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "configs[3] per-track shape: pool of %d tracks, %d tiles of 128 x 128 tracks (%d ordered pairs) per step "
                               "through acx_grid_run, other tiles every step (scores scattered into a device buffer)"
                               % (n, tiles_per_step, int(npairs))},
        "roofline": {"bound": "valu-f64 issue (~ 0.6 - 0.7 busy) next to the LDS permutes of the sliding dot product (0.46 busy); not the scalar cache: one second track for every pair runs at the same rate (scripts/simple_probe2.py)", "kernel": "simple_kernel",
                     "achieved": round(executed / ks / 1e12, 2), "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s",
                     "frac": round(executed / ks / 1e12 / F64_VALU_PEAK_TF, 4), "traffic": _traffic("simple")[0], "traffic_source": _traffic("simple")[1],
                     "kernel_ms_per_step": round(kms, 3), "executed_flop_per_cell": round(30.0 * 64.0 / 54.0, 2),
                     "frac_executed_35_6_flop_per_cell": round(executed / ks / 1e12 / F64_VALU_PEAK_TF, 4),
                     "frac_reference_stomp_54_flop_per_cell": round(54.0 * cells / ks / 1e12 / F64_VALU_PEAK_TF, 4),
                     "frac_survey_8d_model_240_flop_per_cell": round(240.0 * cells / ks / 1e12 / F64_VALU_PEAK_TF, 4),
                     "frac_note": "the 8d model counts a 120-term product per cell that neither the reference's STOMP update nor the "
                                  "kernel performs: its `frac` > 1 says the model over-counts, not that the kernel beats the peak",
                     "reference_stomp_tflops_54_per_cell": round(54.0 * cells / ks / 1e12, 2),
                     "survey_8d_model_tflops_240_per_cell": round(240.0 * cells / ks / 1e12, 2)},
        "cpu_baseline": {"value": round(len(cp) / tcpu, 2), "unit": "track-pairs/s", "cores": 1, "kind": "port",
                         "sample": "first %d pairs of the first timed tile, numpy oracle, BLAS limited to one thread; max relative |diff| vs the "
                                   "GPU's f32 scores %.1e%s" % (len(cp), err, _ref_ratio("simple"))}}


def earlyfusion_leg(ctx, steps=5, warmup=1, n=384, cpu_pairs=32):
    """EarlyFusion per-pair chain (earlyfusion_traile.py:157-198) at 300-500 blocks per track through the pair grid into a
    device buffer: a pool of `n` tracks in 128 x 128 tiles (n = 384: three diagonal tiles of 8128 pairs, three
    off-diagonal ones of 16 384), ONE tile per step, another tile every step."""
    import oracle
    from acoss_amd import _lib, synth
    tracks = synth.earlyfusion_set(n, seed=1, nb_range=(300, 500))
    ctx.ef_upload_pool(tracks)
    nb = np.array([t["mfccs"].shape[0] for t in tracks])
    plan = _lib.grid_plan(nb, _lib.ALGO_EARLYFUSION, True, world=1, tile=128, want_tiles=True)
    tiles = list(plan["tiles"])
    assert len(tiles) >= steps + warmup, len(tiles)
    buf = ctx.dev_alloc(4 * int(plan["floats_per_rank"][0]))
    ep = _lib.EfParams(0.1, 10)
    for s in range(warmup):
        ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr(), first=s, count=1)
    def timed_then_clocked():
        # wall time with the event clocks off (what a caller gets), the kernel times from a second pass over the same tiles
        ctx.profile_enable(False)
        t0 = time.perf_counter()
        for s in range(warmup, warmup + steps):
            ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr(), first=s, count=1)
        dt_ = time.perf_counter() - t0
        ctx.profile_enable(True)
        ctx.profile_reset()
        for s in range(warmup, warmup + steps):
            ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr(), first=s, count=1)
        prof_ = ctx.profile()
        ctx.profile_enable(False)
        return dt_, prof_
    dt, prof = timed_then_clocked()
    host = buf.read(np.float32)
    timed = tiles[warmup:warmup + steps]
    # the same tiles on three bf16 terms per value (round 3's default, all 24 bits of every operand), for comparison
    ctx.set_ef_gemm("bf16x3")
    try:
        ctx.grid_run(plan["spec"], ep, 0, buf.data_ptr(), first=0, count=1)            # (re-splits the pool)
        dt_b, prof_b = timed_then_clocked()
        host_b = buf.read(np.float32)
    finally:
        ctx.set_ef_gemm("default")
    buf.free()
    used = np.zeros(len(host), bool)
    for t in timed:
        used[t.offset:t.offset + 4 * t.rows * t.cols] = True
    same_modes = float(np.mean(host[used] == host_b[used]))
    allp = [p for t in timed for p in _tile_pairs(t)]
    npairs = len(allp)
    pi, pj = np.array(allp).T
    flops = float(np.sum(2.0 * (650 + 1225 + 480) * nb[pi] * nb[pj]))
    g = prof["ef_gemm_kernel"]
    ks = g["ms"] * 1e-3
    t = timed[0]
    cp = _tile_pairs(t)[:cpu_pairs]
    with _one_thread():
        tc = time.perf_counter()
        ref = []
        for a, b in cp:
            sc = oracle.earlyfusion_pair(tracks[a], tracks[b])[0]
            ref.append([sc["mfccs"], sc["ssms"], sc["chromas"], sc["early"]])
        tcpu = time.perf_counter() - tc
    ref = np.array(ref)
    got = np.array([host[t.offset + 4 * ((a - t.row0) * t.cols + (b - t.col0)):][:4] for a, b in cp])
    diff = float(np.max(np.abs(ref - got)))
    same = float(np.mean(np.abs(ref - got) < 1e-4))          # scores are multiples of 0.1: f32 store of the f64 value
    assert diff <= 3.0, diff
    return {
        "metric": "track-pairs/sec, EarlyFusion per-pair chain (3 CSMs, 4 x binarise + Smith-Waterman, kernel fusion) at 300-500 blocks",
        "value": round(npairs / dt, 1), "unit": "track-pairs/s", "n_gpus": 1, "steps": steps,
        "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 3), "higher_is_better": True, "dtype": "f16x2 -> f32",
        "data": "synthetic",
        "config": {"workload": "configs[4] per-track shape: pool of %d tracks of 300-500 blocks, one 128 x 128 grid tile per step "
                               "(diagonal: 8128 pairs, off-diagonal: 16 384), another tile every step, %d pairs in %d steps through "
                               "acx_grid_run (scores scattered into a device buffer)" % (n, npairs, steps)},
        # all three cross-similarity GEMMs run on the 16-bit matrix pipe from two fp16 terms per value: three fp16 products
        # per f32-equivalent multiply-add are what the pipe executes, and what is priced against its dense peak
        "roofline": {"bound": "mfma", "kernel": "ef_gemm_rect_persist_dma_kernel<0> (mfcc / ssm) + <1> (chroma): two fp16 terms per value "
                                                "(ACX_EF_GEMM_F16X2, the default), 256 x 128 tiles over dense rectangles of pairs, one persistent "
                                                "workgroup per CU, operands by LDS-DMA into three buffers (round 6)",
                     "achieved": round(3.0 * flops / ks / 1e12, 1), "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                     "frac": round(3.0 * flops / ks / 1e12 / BF16_MFMA_PEAK_TF, 4), "traffic": _traffic("earlyfusion")[0], "traffic_source": _traffic("earlyfusion")[1],
                     "flops": "executed fp16 flops = 3 x (x1 y2 + x2 y1 + x1 y1) the f32-equivalent 2 (650 + 1225 + 480) nb1 nb2 per pair (SURVEY 8d); dense fp16 peak = dense bf16 peak",
                     "f32_equivalent_tflops": round(flops / ks / 1e12, 2), "f32_mfma_peak_tflops": F32_MFMA_PEAK_TF,
                     "kernel_ms_per_step": round(g["ms"] / steps, 3),
                     "kernels_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in prof.items() if v["launches"]}},
        "bf16x3": {"value": round(npairs / dt_b, 1), "gemm_ms_per_step": round(prof_b["ef_gemm_kernel"]["ms"] / steps, 3),
                   "executed_tflops": round(6.0 * flops / (prof_b["ef_gemm_kernel"]["ms"] * 1e-3) / 1e12, 1),
                   "scores_identical_to_default_fraction": round(same_modes, 6),
                   "note": "ACX_EF_GEMM_BF16X3 on the same tiles: three bf16 terms per value (all 24 bits), six MFMAs per cell, "
                           "ef_gemm_rect_bf16x3_kernel<., 0> (one workgroup per tile, staging registers: its persistent and DMA builds measure "
                           "the same, profiles/r06_ef.md)"},
        "cpu_baseline": {"value": round(len(cp) / tcpu, 3), "unit": "track-pairs/s", "cores": 1, "kind": "port",
                         "sample": "first %d pairs of the first timed tile, numpy + C oracle, BLAS limited to one thread, %.1f s; "
                                   "%.4f of the 4 x %d scores identical to the GPU's, max |diff| %.3g%s"
                                   % (len(cp), tcpu, same, len(cp), diff, _ref_ratio("earlyfusion"))}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-pairs", type=int, default=32)
    args = ap.parse_args()
    from acoss_amd import _lib
    ctx = _lib.Context(0)
    print(json.dumps(serra09_covers_leg(ctx, args.steps, args.warmup)), flush=True)
    print(json.dumps(serra09_f16x2_leg(ctx)), flush=True)
    print(json.dumps(chenfusion_leg(ctx)), flush=True)
    print(json.dumps(simple_leg(ctx, args.steps, args.warmup)), flush=True)
    print(json.dumps(earlyfusion_leg(ctx, args.steps, args.warmup, cpu_pairs=args.cpu_pairs)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
