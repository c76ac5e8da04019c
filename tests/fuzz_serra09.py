"""Test aid (not collected by pytest; needs a GPU): randomised GPU-vs-oracle comparison of Serra09 scores (sizes, parameters, ties).
usage: python tests/fuzz_serra09.py [seconds] [seed]"""
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
t_end = time.time() + budget
rounds = pairs_done = fails = 0
while time.time() < t_end:
    rounds += 1
    m = int(rng.choice([1, 2, 4, 7, 9, 9, 9, 12, 16]))
    kw = dict(m=m, kappa=float(rng.choice([0.0, 0.004, 0.02, 0.095, 0.095, 0.3, 0.7, 1.0])), pct_mode=int(rng.integers(0, 4)),
              inclusive=int(rng.integers(0, 2)), dp_start=int(rng.choice([2, 3])), embed_full=int(rng.integers(0, 2)),
              oti=bool(rng.integers(0, 2)), oti_target=int(rng.integers(0, 2)), dmax=int(rng.integers(0, 2)))
    if rng.random() < 0.3:
        kw.update(gamma_o=float(rng.choice([0.25, 1.0, 1.5])), gamma_e=float(rng.choice([0.25, 0.5, 2.0])))
    ntr = int(rng.integers(3, 7))
    tmax = int(rng.choice([60, 150, 400, 700, 1100, 2040 - m]))
    if tmax > 1500:
        ntr = 3
    tracks = []
    for _ in range(ntr):
        T = int(rng.integers(m + 2, tmax + m + 2))
        kind = rng.random()
        if kind < 0.1:
            T = min(T, 2050 - 2 * m) if tmax > 1500 else T
            x = np.tile(rng.random((1, 12)), (T, 1))       # a constant track: every distance equal
        elif kind < 0.6:
            x = rng.random((T, 12))
        elif kind < 0.85:                      # piecewise constant: heavy ties
            protos = rng.random((int(rng.integers(1, 5)), 12))
            seg = int(rng.integers(1, 30))
            x = protos[np.repeat(rng.integers(0, len(protos), T // seg + 1), seg)[:T]]
        else:                                  # sparse frames with exact zeros
            x = rng.random((T, 12)) * (rng.random((T, 12)) < 0.4)
            x[:, 0] += 0.01
        tracks.append(synth._frame_max_normalise(x))
    frames, offsets = synth.pack(tracks)
    ctx.upload_pool(frames, offsets)
    i, j = np.nonzero(~np.eye(ntr, dtype=bool))
    pairs = np.stack([i, j], 1).astype(np.int32)
    got = ctx.serra09_pairs(pairs, _lib.serra09_params(**kw))
    ref = oracle.serra09_pairs(frames, offsets, pairs, oracle.serra09_params(**kw))
    pairs_done += len(pairs)
    if not np.array_equal(got, ref):
        bad = np.nonzero(got != ref)[0]
        print("MISMATCH round %d kw=%s lens=%s pairs=%s got=%s ref=%s" % (
            rounds, kw, np.diff(offsets).tolist(), pairs[bad[:4]].tolist(), got[bad[:4]], ref[bad[:4]]))
        fails += 1
        if fails >= 8:
            sys.exit(1)
print("fuzz: %d rounds, %d pairs, %d mismatching rounds" % (rounds, pairs_done, fails))
sys.exit(1 if fails else 0)
