"""
GPU tests of the EarlyFusion block-feature kernels (run with -m gpu): acx_ef_block_features /
acx_ef_upload_raw_pool against the oracle's restatement of EarlyFusion.load_features
(oracle.ef_block_features; earlyfusion_traile.py:100-140, 214-247).  f64 arithmetic on both sides,
f32 results: tolerance 2e-6 absolute on the unit-norm MFCC rows and chroma blocks, 1e-5 on the SSM
distances (sqrt of a cancellation).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from acoss_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _track(rng, T, nbeats, nan=False, ncoef=13):
    hpcp = rng.random((T, 12)).astype(np.float32)
    mfcc = rng.standard_normal((T, ncoef)).astype(np.float32)
    if nan:
        mfcc[T // 3, 3] = np.nan
    on = np.sort(rng.choice(T - 10, nbeats, replace=False)).astype(np.int64)
    return dict(chroma=hpcp, mfcc=mfcc, onsets=on)


def _check(got, want):
    assert got["mfccs"].shape == want["mfccs"].shape and got["mfccs"].dtype == np.float32
    np.testing.assert_allclose(got["mfccs"], want["mfccs"], atol=2e-6)
    np.testing.assert_allclose(got["ssms"], want["ssms"], atol=1e-5)
    np.testing.assert_allclose(got["chromas"], want["chromas"], atol=2e-6)
    np.testing.assert_array_equal(got["chroma_med"], np.asarray(want["chroma_med"], np.float64))


def test_block_features_against_oracle(ctx):
    import oracle
    rng = np.random.default_rng(11)
    cases = [_track(rng, 2600, 48, nan=True),                 # ~ 55 frames per beat: 20-beat blocks of ~ 1000 frames (20 : 1)
             _track(rng, 700, 60),                            # ~ 12 frames per beat: blocks of ~ 230 frames
             _track(rng, 140, 50),                            # blocks shorter than 50 rows: upsampling, no blur
             _track(rng, 3001, 21),                           # a single block; odd number of frames (median of the middle)
             _track(rng, 500, 20), _track(rng, 400, 7)]       # no block at all
    for t in cases:
        got = ctx.ef_block_features(t["chroma"], t["mfcc"], t["onsets"])
        want = oracle.ef_block_features(t["chroma"], t["mfcc"], t["onsets"])
        _check(got, want)
    # other block geometry
    t = _track(rng, 1500, 40, ncoef=20)
    got = ctx.ef_block_features(t["chroma"], t["mfcc"], t["onsets"], blocksize=12, mfccs_per_block=32, chromas_per_block=24)
    want = oracle.ef_block_features(t["chroma"], t["mfcc"], t["onsets"], 12, 32, 24)
    _check(got, want)
    # beats that repeat / run past the end of the features: numpy slicing clips
    t = _track(rng, 900, 40)
    t["onsets"][-3:] = [880, 950, 2000]
    t["onsets"][5] = t["onsets"][4]
    _check(ctx.ef_block_features(t["chroma"], t["mfcc"], t["onsets"]), oracle.ef_block_features(t["chroma"], t["mfcc"], t["onsets"]))
    with pytest.raises(NotImplementedError):
        ctx.ef_block_features(t["chroma"], t["mfcc"], t["onsets"], mfccs_per_block=65)


def test_block_features_against_reference_with_skimage(ctx, golden):
    """The device against the reference itself run with the real scikit-image (tests/golden/efprep_skimage.npz,
    made by tests/golden/make_efprep_goldens.py): f64 on both sides, f32 results, 2e-6 / 1e-5 like the oracle test."""
    g = golden("efprep_skimage")
    for k in (0, 1, 2, 9):
        args = dict(blocksize=20, mfccs_per_block=50, chromas_per_block=40) if k != 9 else dict(blocksize=12, mfccs_per_block=32, chromas_per_block=24)
        got = ctx.ef_block_features(g["lf%d_hpcp" % k], np.ascontiguousarray(g["lf%d_mfcc_htk" % k].T), g["lf%d_onsets" % k], **args)
        want = {key: g["lf%d_%s" % (k, key)] for key in ("mfccs", "ssms", "chromas", "chroma_med")}
        np.testing.assert_allclose(got["mfccs"], want["mfccs"], atol=2e-6)
        np.testing.assert_allclose(got["ssms"], want["ssms"], atol=1e-5)
        np.testing.assert_allclose(got["chromas"], want["chromas"], atol=2e-6)
        np.testing.assert_array_equal(got["chroma_med"].astype(np.float32), want["chroma_med"])
    # resize_block alone through a one-block track: chroma rows i1 .. i2 -> 40 rows, incl. the case skimage's clip decides
    C = g["rb_C"]
    for k, (which, i1, i2, rows) in enumerate(g["rb_cases"]):
        if which != 1 or rows != 40:
            continue
        on = np.concatenate([[i1], np.full(19, i1, np.int64), [i2]]).astype(np.int64)      # 21 beats -> one block [i1, i2)
        mf = np.zeros((len(C), 13), np.float32)
        got = ctx.ef_block_features(C, mf, on)
        np.testing.assert_allclose(got["chromas"][0].reshape(40, 12), g["rb_out_%d" % k], atol=2e-6, err_msg="case %d" % k)


def test_raw_pool_equals_uploaded_block_features(ctx):
    """The collection-level entry point keeps the features on the device: the pair scores equal those
    of a pool uploaded from the per-track results (same kernels, same bits), and the oracle's
    features give the same scores up to the binarisation ties a 1e-7 perturbation can move."""
    import oracle
    rng = np.random.default_rng(5)
    tracks = [_track(rng, int(rng.integers(1500, 2600)), int(rng.integers(60, 90))) for _ in range(5)]
    boff = ctx.ef_upload_raw_pool(tracks)
    assert np.array_equal(np.diff(boff), [len(t["onsets"]) - 20 for t in tracks])
    pairs = oracle.all_pairs(5, True).astype(np.int32)
    a = ctx.earlyfusion_pairs(pairs)
    feats = [ctx.ef_block_features(t["chroma"], t["mfcc"], t["onsets"]) for t in tracks]
    ctx.ef_upload_pool(feats)
    b = ctx.earlyfusion_pairs(pairs)
    assert np.array_equal(a, b)
    ctx.ef_upload_pool([oracle.ef_block_features(t["chroma"], t["mfcc"], t["onsets"]) for t in tracks])
    c = ctx.earlyfusion_pairs(pairs)
    assert np.max(np.abs(a - c)) <= 2.0


def test_earlyfusion_class_from_raw_feature_files(tmp_path, monkeypatch):
    """EarlyFusion over track files that hold raw features only: load_features(i) computes the blocks
    on the device, all_pairwise builds the whole pool there, and the grid equals the pair list."""
    import oracle
    from acoss_amd.algorithms.earlyfusion_traile import EarlyFusion
    from acoss_amd.featurestore import save_track
    rng = np.random.default_rng(2)
    labels = ["a", "a", "b", "b", "c"]
    root = str(tmp_path) + "/feat/"
    raw = []
    with open(tmp_path / "ds.csv", "w") as f:
        f.write("work_id,track_id\n")
        for k, l in enumerate(labels):
            t = _track(rng, 1800 + 100 * k, 50 + 3 * k)
            raw.append(t)
            save_track(root + "%s/t%d.h5" % (l, k), {"label": l, "track_id": "t%d" % k, "hpcp": t["chroma"],
                                                     "mfcc_htk": t["mfcc"].T, "madmom_features": {"onsets": t["onsets"]}})
            f.write("%s,t%d\n" % (l, k))
    monkeypatch.chdir(tmp_path)
    ef = EarlyFusion(str(tmp_path / "ds.csv"), root, chroma_type="hpcp", shortname="toy", log_times=True)
    got = ef.load_features(1)
    want = oracle.ef_block_features(raw[1]["chroma"], raw[1]["mfcc"], raw[1]["onsets"])
    _check(got, want)
    assert ef.load_features(1) is got and len(ef.times["features"]) == 1
    ef2 = EarlyFusion(str(tmp_path / "ds.csv"), root, chroma_type="hpcp", shortname="toy2")
    ef2.all_pairwise(symmetric=True)
    pairs = oracle.all_pairs(5, True).astype(np.int32)
    sc = ef2._ctx.earlyfusion_pairs(pairs, kappa=ef2.kappa, K=ef2.K)
    for c, s in enumerate(("mfccs", "ssms", "chromas", "early")):
        D = np.zeros((5, 5), np.float32)
        D[pairs[:, 0], pairs[:, 1]] = sc[:, c]
        D += D.T
        assert np.array_equal(np.array(ef2.Ds[s]), D)
    assert ef2.cliques == {"a": {0, 1}, "b": {2, 3}, "c": {4}}
