"""Test aid (not collected by pytest; needs a GPU): randomised GPU-vs-oracle comparison of the SiMPle, Smith-Waterman, EarlyFusion and
ChenFusion entry points.   usage: python tests/fuzz_other.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402
from acoss_amd import _lib, synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
ctx = _lib.Context(0)
fails = 0


def report(what, *info):
    global fails
    fails += 1
    print("MISMATCH", what, *info)


def fuzz_simple():
    n = int(rng.integers(2, 6))
    feats = []
    for _ in range(n):
        L = int(rng.choice([10, 11, 13, 30, 64, 65, 130, 257, 400]))
        f = rng.random((12, L))
        if rng.random() < 0.3:
            f = np.repeat(f[:, : L // 7 + 1], 7, axis=1)[:, :L] + 1e-3 * rng.random((12, L))
        feats.append(oracle.simple_smooth(f))
    tracks = [np.ascontiguousarray(f.T, dtype=np.float64) for f in feats]
    offs = np.concatenate([[0], np.cumsum([len(t) for t in tracks])]).astype(np.int64)
    ctx.upload_pool_f64(np.concatenate(tracks), offs)
    i, j = np.nonzero(~np.eye(n, dtype=bool))
    pairs = np.stack([i, j], 1).astype(np.int32)
    L = int(rng.choice([2, 4, 10, 10, 10]))
    L = min(L, min(f.shape[1] for f in feats) - 1)
    got = ctx.simple_pairs(pairs, L)
    ref = np.array([-oracle.simple_sim(feats[a], oracle.simple_oti(feats[a], feats[b])[0], L) for a, b in pairs])
    if not np.allclose(got, ref, rtol=1e-11, atol=1e-13):
        report("simple", [f.shape[1] for f in feats], L, np.max(np.abs(got - ref)))
    return len(pairs)


def fuzz_sw():
    m, n = int(rng.integers(3, 513)), int(rng.integers(3, 513))
    dens = float(rng.choice([0.02, 0.1, 0.3, 0.6, 0.95]))
    B = (rng.random((m, n)) < dens).astype(np.uint8)
    if rng.random() < 0.3:
        k = min(m, n)
        B[np.arange(k), np.arange(k)] = 1
    if round(ctx.sw_binary(B) * 10) != oracle.sw_constrained_i32(B):
        report("sw", m, n, dens)
    return 1


def fuzz_ef():
    n = int(rng.integers(2, 5))
    lo = int(rng.choice([12, 20, 40, 90]))
    tracks = synth.earlyfusion_set(n, seed=int(rng.integers(1 << 30)), nb_range=(lo, lo + int(rng.integers(1, 120))))
    if rng.random() < 0.5:
        for key in ("mfccs", "ssms", "chromas"):
            k = min(len(tracks[0][key]), len(tracks[1][key])) - 2
            tracks[1][key][1:1 + k] = tracks[0][key][:k] + 0.02 * rng.standard_normal(tracks[0][key][:k].shape).astype(np.float32)
    ctx.ef_upload_pool(tracks)
    kappa = float(rng.choice([0.05, 0.1, 0.1, 0.25]))
    K = int(rng.choice([3, 10, 10, 11]))
    cnt = 0
    for (i, j) in [(0, 1), (1, 0)] + ([(n - 1, 0)] if n > 2 else []):
        d = ctx.ef_debug_pair(i, j, kappa, K)
        if d["oti"] != oracle.get_oti(tracks[i]["chroma_med"], tracks[j]["chroma_med"]):
            report("ef oti", i, j)
        for k in range(3):
            if abs(oracle.sw_constrained(oracle.csm_to_binary(d["csm"][k], kappa)) - float(d["scores"][k])) > 1e-5:
                report("ef score given device csm", k, d["csm"][k].shape, kappa)
        if abs(oracle.sw_constrained(oracle.csm_to_binary(d["fused"], kappa)) - float(d["scores"][3])) > 1e-5:
            report("ef early score given device fused", d["fused"].shape, kappa, K)
        ws = np.zeros_like(d["csm"][0])
        for k in range(3):
            ws += oracle.get_wcsm(d["csm"][k], K, K)
        if not np.allclose(d["fused"], np.exp(-ws), rtol=2e-3, atol=1e-6):
            report("ef fused", d["fused"].shape, K)
        got = ctx.earlyfusion_pairs(np.array([[i, j]], np.int32), kappa, K)
        if not np.array_equal(got[0], d["scores"]):
            report("ef batch vs debug", got[0], d["scores"])
        cnt += 1
    return cnt


def fuzz_chen():
    n = int(rng.integers(2, 5))
    tracks = [synth._frame_max_normalise(rng.random((int(rng.integers(12, 500)), 12))) for _ in range(n)]
    frames, offsets = synth.pack(tracks)
    ctx.upload_pool(frames, offsets)
    i, j = np.nonzero(~np.eye(n, dtype=bool))
    pairs = np.stack([i, j], 1).astype(np.int32)
    kw = dict(kappa=float(rng.choice([0.095, 0.3])), dp_start=int(rng.choice([2, 3])))
    if rng.random() < 0.5:
        kw.update(gamma_o=float(rng.choice([0.25, 1.0])), gamma_e=float(rng.choice([0.5, 2.0])))
    got = ctx.chenfusion_pairs(pairs, _lib.serra09_params(**kw))
    q = oracle.serra09_pairs(frames, offsets, pairs, oracle.serra09_params(dmax=0, **kw))
    dm = oracle.serra09_pairs(frames, offsets, pairs, oracle.serra09_params(dmax=1, **kw))
    if not (np.array_equal(got[:, 0], q) and np.array_equal(got[:, 1], dm)):
        report("chenfusion", kw, np.diff(offsets).tolist())
    return len(pairs)


counts = {}
t_end = time.time() + budget
legs = [("simple", fuzz_simple), ("sw", fuzz_sw), ("ef", fuzz_ef), ("chen", fuzz_chen)]
while time.time() < t_end and fails < 8:
    name, fn = legs[int(rng.integers(0, len(legs)))]
    counts[name] = counts.get(name, 0) + fn()
print("fuzz_other:", counts, "mismatches:", fails)
sys.exit(1 if fails else 0)
