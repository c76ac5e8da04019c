"""
GPU parity tests for SiMPle: the HIP kernel (f64) through the C ABI against the oracle and
against the golden vectors captured from the reference's own Simple class
(tests/golden/simple.npz).  Tolerance: 1e-11 relative (f64 on both sides; only the order of
the 120-term dot products differs).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-11


@pytest.fixture(scope="module")
def ctx():
    from acoss_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _pool(feats):
    tracks = [np.ascontiguousarray(f.T, dtype=np.float64) for f in feats]
    offs = np.concatenate([[0], np.cumsum([len(t) for t in tracks])]).astype(np.int64)
    return np.concatenate(tracks), offs


def test_reference_goldens(ctx, golden):
    g = golden("simple")
    for k in range(6):
        A, B = g["sim_A_%d" % k], g["sim_B_%d" % k]
        fr, offs = _pool([A, B])
        ctx.upload_pool_f64(fr, offs)
        got = ctx.simple_pairs(np.array([[0, 1]], np.int32), 10)[0]
        ref = -float(g["sim_out_%d" % k])          # Simple.similarity stores -simple_sim(Si, oti(Si, Sj))
        assert abs(got - ref) <= RTOL * max(1.0, abs(ref)), (k, got, ref)
        # without OTI on the already-rolled B: simple_sim alone
        fr, offs = _pool([A, g["oti_B_%d" % k]])
        ctx.upload_pool_f64(fr, offs)
        got2 = ctx.simple_pairs(np.array([[0, 1]], np.int32), 10, oti=False)[0]
        assert abs(got2 - ref) <= RTOL * max(1.0, abs(ref))


def test_all_ordered_pairs_against_oracle(ctx):
    import oracle
    from acoss_amd import synth
    d = synth.cover_set(n_works=4, versions=3, seed=12, t_range=(2500, 9000))
    n = len(d["offsets"]) - 1
    feats = [oracle.simple_features(d["frames"][d["offsets"][i]:d["offsets"][i + 1]]) for i in range(n)]
    assert min(f.shape[1] for f in feats) >= 25 and max(f.shape[1] for f in feats) <= 90
    fr, offs = _pool(feats)
    ctx.upload_pool_f64(fr, offs)
    pairs = oracle.all_pairs(n, False).astype(np.int32)
    got = ctx.simple_pairs(pairs, 10)
    ref = np.array([oracle.simple_pair(feats[i], feats[j]) for i, j in pairs])
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=1e-13)
    assert not np.allclose(got.reshape(-1), got.reshape(-1)[::-1])      # ordered: D is not symmetric
    for L in (4, 16):
        got = ctx.simple_pairs(pairs[:20], L)
        ref = np.array([-oracle.simple_sim(feats[i], oracle.simple_oti(feats[i], feats[j])[0], L) for i, j in pairs[:20]])
        np.testing.assert_allclose(got, ref, rtol=RTOL, atol=1e-13)


def test_large_and_ragged_tracks(ctx):
    import oracle
    rng = np.random.default_rng(4)
    S = type("S", (), {})
    feats = [oracle.simple_smooth(rng.random((12, n))) for n in (10, 11, 64, 257, 512, 73, 74, 137, 1500, 1203)]
    fr, offs = _pool(feats)
    ctx.upload_pool_f64(fr, offs)
    pairs = np.array([[0, 1], [1, 0], [0, 4], [4, 0], [2, 3], [3, 4], [4, 3], [5, 6], [6, 5], [7, 2], [2, 7], [5, 5],
                      [8, 9], [9, 8], [8, 0], [0, 8], [3, 8]], np.int32)
    got = ctx.simple_pairs(pairs, 10)
    ref = np.array([oracle.simple_pair(feats[i], feats[j]) for i, j in pairs])
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=1e-13)


def test_errors(ctx):
    from acoss_amd import _lib
    rng = np.random.default_rng(1)
    fr, offs = _pool([rng.random((12, 9)), rng.random((12, 30)), rng.random((12, 6001))])
    ctx.upload_pool_f64(fr, offs)
    with pytest.raises(_lib.AcxError):
        ctx.simple_pairs(np.array([[0, 1]], np.int32), 10)          # shorter than SSLEN
    with pytest.raises(NotImplementedError):
        ctx.simple_pairs(np.array([[1, 2]], np.int32), 10)          # > 6000 pooled frames (10 hours of audio)
    with pytest.raises(ValueError):
        ctx.simple_pairs(np.array([[1, 5]], np.int32), 10)


def test_simple_class_end_to_end(tmp_path, monkeypatch):
    """benchmark(algorithm="SiMPle") over feature files == oracle on the same data."""
    import acoss_amd
    import oracle
    from acoss_amd import synth
    from acoss_amd.featurestore import save_track
    d = synth.cover_set(clique_sizes=[2, 3, 2, 1], seed=3, t_range=(2600, 6000))
    n = len(d["offsets"]) - 1
    root = str(tmp_path) + "/feat/"
    with open(tmp_path / "ds.csv", "w") as f:
        f.write("work_id,track_id\n")
        for i in range(n):
            save_track(root + "%s/t%d.h5" % (d["labels"][i], i),
                       {"label": d["labels"][i], "track_id": "t%d" % i,
                        "hpcp": d["frames"][d["offsets"][i]:d["offsets"][i + 1]]})
            f.write("%s,t%d\n" % (d["labels"][i], i))
    monkeypatch.chdir(tmp_path)
    res = acoss_amd.benchmark(str(tmp_path / "ds.csv"), root, feature_type="hpcp", algorithm="SiMPle", shortname="toy")
    feats = [oracle.simple_features(d["frames"][d["offsets"][i]:d["offsets"][i + 1]]) for i in range(n)]
    D = np.zeros((n, n), np.float32)
    for i, j in oracle.all_pairs(n, False):
        D[i, j] = oracle.simple_pair(feats[i], feats[j])
    cl = {}
    for i, l in enumerate(d["labels"]):
        cl.setdefault(l, []).append(i)
    want = oracle.eval_statistics(D, list(cl.values()))
    np.testing.assert_allclose(np.array(res["main"][:4]), np.array(want[:4]), rtol=1e-9)
