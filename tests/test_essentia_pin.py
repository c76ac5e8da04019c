"""
Essentia pin kit, part 2: consumes tests/golden/serra09_essentia.npz when it exists (made by
tests/golden/make_serra09_essentia_goldens.py on a machine with essentia) and holds the oracle -- and,
with -m gpu, the device -- to essentia's own outputs for the calls acoss makes
(acoss/algorithms/rqa_serra09.py:60-67, latefusion_chen.py:63-72).

Without the file every test here is skipped: the Serra09 chain stays "parity unpinned" (DESIGN.md
section 2).  With it, the tests search the switches that only exist because essentia's details
were recalled, not read (embed_full, pct_mode, oti_target, dp_start, inclusive, arith), report which
combinations reproduce essentia's cross-similarity matrix and distances on EVERY stored pair, and
fail unless the library's DEFAULT combination is one of them.
"""
import itertools
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "serra09_essentia.npz")

needs_file = pytest.mark.skipif(not os.path.exists(PATH), reason="no essentia fixture (tests/golden/serra09_essentia.npz): "
                                "run tests/golden/make_serra09_essentia_goldens.py where essentia is installed")

SWITCHES = dict(embed_full=(0, 1), pct_mode=(0, 1, 2, 3), oti_target=(0, 1), dp_start=(2, 3), inclusive=(1, 0))
DEFAULT = dict(embed_full=0, pct_mode=0, oti_target=0, dp_start=2, inclusive=1)


def _load():
    z = np.load(PATH, allow_pickle=False)
    tracks = [z["track_%d" % i] for i in range(int(z["n_tracks"]))]
    return z, tracks, [tuple(p) for p in z["pairs"]]


def _combos():
    keys = list(SWITCHES)
    for vals in itertools.product(*[SWITCHES[k] for k in keys]):
        yield dict(zip(keys, vals))


def _orient(R, csm):
    """essentia's matrix in the oracle's orientation (rows = query frames), or None."""
    if csm.shape == R.shape:
        return csm
    if csm.T.shape == R.shape:
        return csm.T
    return None


def _search(run_pair):
    """run_pair(case params, switches, q, r, align) -> (R uint8 (Mq, Mr), score).  Returns the list of
    switch combinations that reproduce essentia everywhere, and a per-combination miss report."""
    z, tracks, pairs = _load()
    winners, report = [], {}
    for sw in _combos():
        misses = 0
        for name in [str(c) for c in z["cases"]]:
            m, tau, kappa, oti = z["case_%s" % name]
            case = dict(m=int(m), tau=int(tau), kappa=float(kappa), oti=bool(oti))
            for k, (i, j) in enumerate(pairs):
                key = "%s_csm_%d" % (name, k)
                if key not in z.files:
                    continue
                for align in ("serra09", "chen17"):
                    R, score = run_pair(case, sw, tracks[i], tracks[j], align)
                    csm = _orient(R, z[key])
                    if csm is None or not np.array_equal(R, csm):
                        misses += 1
                    elif float(z["%s_%s_distance_%d" % (name, align, k)]) != score:
                        misses += 1
        report[tuple(sorted(sw.items()))] = misses
        if misses == 0:
            winners.append(sw)
    return winners, report


def _verdict(winners, report, who):
    best = sorted(report.items(), key=lambda kv: kv[1])[:5]
    msg = "%s: %d switch combination(s) reproduce essentia on every pair: %s\nclosest: %s" % (
        who, len(winners), winners, [(dict(k), v) for k, v in best])
    print(msg)
    assert winners, msg
    assert DEFAULT in winners, "the DEFAULT switches do not reproduce essentia -- change the defaults of " \
                               "acx_serra09_default_params / oracle.serra09_params to one of: %s" % winners


@needs_file
def test_oracle_against_essentia():
    """Both arithmetics of the oracle are searched; the kernels implement "tree", so that is the one
    whose default switches must win (a win of "seq108" only says: the arithmetic spec itself has to
    move)."""
    import oracle
    results = {}
    for arith in ("tree", "seq108"):
        def run_pair(case, sw, q, r, align, arith=arith):
            p = oracle.serra09_params(arith=arith, dmax=int(align == "chen17"), **case, **sw)
            s, it = oracle.serra09_pair(q, r, p, want_intermediates=True)
            return it["R"], s
        results[arith] = _search(run_pair)
        print("oracle[%s]: winners %s" % (arith, results[arith][0]))
    _verdict(results["tree"][0], results["tree"][1], "oracle[tree] (seq108 winners: %s)" % results["seq108"][0])


@needs_file
@pytest.mark.gpu
def test_device_against_essentia():
    from acoss_amd import _lib, synth
    z, tracks, pairs = _load()
    frames, offsets = synth.pack(tracks)
    ctx = _lib.Context(0)
    try:
        ctx.upload_pool(frames, offsets)
        index = {id(t): i for i, t in enumerate(tracks)}

        def run_pair(case, sw, q, r, align):
            p = _lib.serra09_params(dmax=int(align == "chen17"), **case, **sw)
            g = ctx.serra09_debug_pair(index[id(q)], index[id(r)], p)
            R = ((g["d2"] <= g["thr_q"][:, None]) & (g["d2"] <= g["thr_r"][None, :])).astype(np.uint8)
            return R, g["score"]
        winners, report = _search(run_pair)
        _verdict(winners, report, "device")
    finally:
        ctx.close()


def test_pin_kit_is_wired():
    """Always runs: the generator script exists, names the calls acoss makes, and this module finds
    its output where the script writes it."""
    src = open(os.path.join(HERE, "golden", "make_serra09_essentia_goldens.py")).read()
    for needle in ("ChromaCrossSimilarity", "CoverSongSimilarity", "binarizePercentile", "alignmentType=align",
                   "distanceType=\"symmetric\"", "serra09_essentia.npz"):
        assert needle in src
    assert PATH.endswith(os.path.join("tests", "golden", "serra09_essentia.npz"))
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_serra09_essentia_goldens.py")],
                       capture_output=True, text=True)
    try:
        import essentia  # noqa: F401
    except ImportError:
        assert r.returncode != 0 and "essentia is not importable" in (r.stderr + r.stdout)
