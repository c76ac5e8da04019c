"""
Randomised cross-checks of the EarlyFusion device chain (a development tool like tests/fuzz_serra09.py; `python
tests/fuzz_earlyfusion.py [rounds] [seed]` on the GPU box; tests/test_gpu_earlyfusion.py::test_fuzz_small runs a few rounds).

Every round draws a pool of ragged tracks (1 .. 140 blocks, sometimes one of 520 or 1100), odd feature widths
(mfcc / ssm widths that are not multiples of 32, chroma blocks of 8, 16 or 40 frames), random kappa / K, and a random pair
list (duplicates, self pairs, both orders), and checks
  * the default arithmetic (two fp16 terms per value, rectangles, bits path) against the exact-f32 one-matrix-at-a-time
    kernels: mfccs / ssms / chromas / early scores equal except on ties (>= 92 % identical, none further than 3.0);
  * a sample of pairs against the numpy oracle within the same bound;
  * the pair grid against the pair list.
"""
import sys

import numpy as np


def make_case(rng):
    n = int(rng.integers(3, 40))
    G = int(rng.choice([8, 16, 40]))
    d0, d1 = int(rng.integers(20, 700)), int(rng.integers(20, 1300))
    nbs = [int(v) for v in rng.integers(1, 140, n)]
    if rng.random() < 0.3:
        nbs[int(rng.integers(0, n))] = int(rng.choice([520, 1100]))

    def track(nb):
        return dict(mfccs=rng.standard_normal((nb, d0)).astype(np.float32), ssms=(2 * rng.random((nb, d1))).astype(np.float32),
                    chromas=(rng.random((nb, 12 * G)).astype(np.float32) ** 2), chroma_med=rng.random(12) ** 2)
    tracks = [track(nb) for nb in nbs]
    # value ranges the per-row scales of the f16x2 operands have to follow: whole tracks scaled by up to 10^+-3, single
    # elements 100 times their row's size, rows of zeros
    ill = False                 # scaled tracks / outliers / zero rows: ill-conditioned ON PURPOSE (see check_case)
    if rng.random() < 0.3:
        ill = True
        for t in tracks:
            t["mfccs"] *= np.float32(10.0 ** rng.uniform(-3, 3))
            t["ssms"] *= np.float32(10.0 ** rng.uniform(-3, 3))
    if rng.random() < 0.3:
        ill = True
        for t in tracks:
            nb = t["mfccs"].shape[0]
            for key in ("mfccs", "ssms"):
                r = rng.integers(0, nb, 3)
                cidx = rng.integers(0, t[key].shape[1], 3)
                t[key][r, cidx] *= np.float32(100.0)
            if nb > 6 and rng.random() < 0.3:
                t["mfccs"][int(rng.integers(0, nb))] = 0.0
    K = int(rng.choice([1, 3, 10, 16, 17]))
    kappa = float(rng.choice([0.05, 0.1, 0.3]))
    m = int(rng.integers(1, 300))
    pairs = rng.integers(0, n, (m, 2)).astype(np.int32)
    # the reference's getWCSM needs more than K blocks per track (np.partition): keep to such pairs
    ok = np.array([min(nbs[a], nbs[b]) > K and min(nbs[a], nbs[b]) >= 4 for a, b in pairs])
    return dict(n=n, G=G, d0=d0, d1=d1, nbs=nbs, tracks=tracks, K=K, kappa=kappa, pairs=np.ascontiguousarray(pairs[ok]),
                grid_tile=int(rng.choice([0, 5])), sample=rng.integers(0, 1 << 30, 3), ill=ill)


def check_case(ctx, c, oracle, _lib):
    n, nbs, tracks, K, kappa, pairs = c["n"], c["nbs"], c["tracks"], c["K"], c["kappa"], c["pairs"]
    tag = (n, c["G"], c["d0"], c["d1"], K, kappa)
    ctx.ef_upload_pool(tracks)
    if len(pairs) == 0:
        return 0
    ctx.set_ef_gemm("default")
    got = ctx.earlyfusion_pairs(pairs, kappa=kappa, K=K)
    ctx.set_ef_gemm("f32")
    want = ctx.earlyfusion_pairs(pairs, kappa=kappa, K=K)
    ctx.set_ef_gemm("default")
    # (a track against itself has an exactly-zero diagonal in one arithmetic and rounding noise in the other; with K = 1 the
    # reference's own getWCSM divides 0 by 0 there: self pairs stay in the list, but are not compared)
    other = pairs[:, 0] != pairs[:, 1]
    _, first = np.unique(pairs[:, 0].astype(np.int64) * n + pairs[:, 1], return_index=True)      # (a repeated pair counts once)
    uniq = np.zeros(len(pairs), bool)
    uniq[first] = True
    other &= uniq
    same = np.all(got == want, axis=1)[other]
    if other.any():
        assert np.max(np.abs(got - want)[other]) <= 3.0, (tag, float(np.max(np.abs(got - want)[other])))
        # (i.i.d. random features at kappa = 0.3 put many cells on a threshold: a few per cent of the scores move by tenths)
        assert (~same).sum() <= max(3, int(0.08 * len(same))), (tag, float(same.mean()))
    # a few pairs against the oracle
    for k in c["sample"] % len(pairs):
        a, b = int(pairs[k, 0]), int(pairs[k, 1])
        if max(nbs[a], nbs[b]) > 300 or a == b:
            continue
        sc = oracle.earlyfusion_pair(tracks[a], tracks[b], kappa=kappa, K=K)[0]
        ref = np.array([sc["mfccs"], sc["ssms"], sc["chromas"], sc["early"]])
        if not c["ill"]:
            # well-conditioned rounds: the plain bar, 3.0 against the oracle (the widened bound below is for the
            # deliberately ill-conditioned cases only)
            assert np.all(np.abs(got[k] - ref) <= 3.0), (tag, a, b, got[k], ref)
            continue
        # The scaled / outlier cases are ill-conditioned on purpose (a row with one value 100 times the others is equally far
        # from everything: its ranks hang on the last bits of ANY arithmetic), so the bar is the spread the reference's own
        # arithmetic shows between f32- and f64-evaluated matrices, + 3.0 -- not 3.0 against one of them
        s64 = oracle.earlyfusion_pair(tracks[a], tracks[b], kappa=kappa, K=K, csm_f64=True)[0]
        ref64 = np.array([s64["mfccs"], s64["ssms"], s64["chromas"], s64["early"]])
        off = np.minimum(np.abs(got[k] - ref), np.abs(got[k] - ref64))
        assert np.all(off <= 3.0 + np.abs(ref - ref64)), (tag, a, b, got[k], ref, ref64)
    # the grid against the list
    if n <= 24 and min(nbs) > K and min(nbs) >= 4:
        planes = [np.zeros((n, n), np.float32) for _ in range(4)]
        ctx.pair_grid(_lib.ALGO_EARLYFUSION, True, _lib.EfParams(kappa, K), planes, mirror=False, tile=c["grid_tile"])
        iu, ju = np.triu_indices(n, 1)
        lst = ctx.earlyfusion_pairs(np.stack([iu, ju], 1).astype(np.int32), kappa=kappa, K=K)
        for e in range(4):
            assert np.array_equal(planes[e][iu, ju], lst[:, e]), (tag, e)
    return len(pairs)


def one_round(ctx, rng, oracle, _lib):
    return check_case(ctx, make_case(rng), oracle, _lib)


def run(rounds=20, seed=0, ctx=None):
    import oracle
    from acoss_amd import _lib
    own = ctx is None
    if own:
        ctx = _lib.Context(0)
    rng = np.random.default_rng(seed)
    total = 0
    try:
        for _ in range(rounds):
            total += one_round(ctx, rng, oracle, _lib)
    finally:
        ctx.set_ef_gemm("default")
        if own:
            ctx.close()
    return total


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("pairs checked:", run(r, s))
