#!/usr/bin/env python3
"""
Essentia pin kit, part 1: produce tests/golden/serra09_essentia.npz on a machine that HAS essentia.

The Serra09 / ChenFusion arithmetic of the reference lives in essentia (un-pinned dependency,
/root/reference/setup.py:53; call sites acoss/algorithms/rqa_serra09.py:60-67 and
latefusion_chen.py:63-72), which is absent from the build container and from the GPU box, so the
oracle's Serra09 chain is "parity unpinned".  This script is the one command that pins it:

    python tests/golden/make_serra09_essentia_goldens.py          # needs `import essentia`

It feeds seeded synthetic chroma (acoss_amd.synth; no reference code involved) through exactly the
calls acoss makes --

    ChromaCrossSimilarity(frameStackSize=m, frameStackStride=tau, binarizePercentile=kappa, oti=oti)
    CoverSongSimilarity(alignmentType='serra09' | 'chen17', distanceType='symmetric')

-- and stores inputs, the binary cross-similarity matrix, the score matrices and the distances.
tests/test_essentia_pin.py (CPU: the oracle; GPU: the device) picks the file up automatically and
reports which combination of the oracle's switches (embed_full, pct_mode, oti_target, dp_start,
inclusive, arith) reproduces essentia.  The .npz holds data only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = [
    # (name, m, tau, kappa, oti)
    ("default", 9, 1, 0.095, True),
    ("no_oti", 9, 1, 0.095, False),
    ("m4_tau2", 4, 2, 0.095, True),
    ("kappa20", 9, 1, 0.2, True),
]


def inputs():
    from acoss_amd import synth
    rng = np.random.default_rng(20260101)
    cov = synth.cover_set(clique_sizes=[2, 2], seed=11, t_range=(60, 180))
    tracks = [cov["frames"][cov["offsets"][i]:cov["offsets"][i + 1]] for i in range(4)]
    tracks.append(synth._frame_max_normalise(rng.random((201 + 9, 12))))   # n - 1 = 200: (n-1) * 0.095 = 19 exactly
    tracks.append(synth._frame_max_normalise(rng.random((97, 12))))
    tracks.append(np.repeat(synth._frame_max_normalise(rng.random((6, 12))), 20, axis=0))   # heavy ties
    pairs = [(0, 1), (1, 0), (0, 2), (2, 3), (4, 5), (5, 4), (4, 4), (6, 0), (6, 6)]
    return tracks, pairs


def main():
    try:
        import essentia
        from essentia.standard import ChromaCrossSimilarity, CoverSongSimilarity
    except ImportError:
        sys.exit("essentia is not importable here: run this script on a machine with essentia installed "
                 "(pip install essentia) and commit tests/golden/serra09_essentia.npz")
    tracks, pairs = inputs()
    out = {"essentia_version": np.array(essentia.__version__), "n_tracks": np.array(len(tracks)),
           "pairs": np.array(pairs, np.int32), "cases": np.array([c[0] for c in CASES])}
    for i, t in enumerate(tracks):
        out["track_%d" % i] = np.ascontiguousarray(t, np.float32)
    for name, m, tau, kappa, oti in CASES:
        out["case_%s" % name] = np.array([m, tau, kappa, float(oti)], np.float64)
        for k, (i, j) in enumerate(pairs):
            q, r = tracks[i], tracks[j]
            if min(len(q), len(r)) <= m * tau + 1:
                continue
            csm = ChromaCrossSimilarity(frameStackSize=m, frameStackStride=tau, binarizePercentile=kappa, oti=oti)(q, r)
            csm = np.asarray(csm, np.float32)
            out["%s_csm_%d" % (name, k)] = csm.astype(np.uint8)
            for align in ("serra09", "chen17"):
                smat, dist = CoverSongSimilarity(alignmentType=align, distanceType="symmetric")(csm)
                out["%s_%s_scorematrix_%d" % (name, align, k)] = np.asarray(smat, np.float32)
                out["%s_%s_distance_%d" % (name, align, k)] = np.array(dist, np.float32)
    path = os.path.join(HERE, "serra09_essentia.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d arrays, essentia %s)" % (path, len(out), essentia.__version__))


if __name__ == "__main__":
    main()
