"""
GPU parity test for the similarity-network-fusion loop (acx_snf_fuse) through the C ABI: against the oracle's
restatement of the reference (oracle.snf_fuse, pinned to the reference's own output by
tests/golden).  f64 on both sides; only the order of the K-term and row sums differs: rtol 1e-10.
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


def _scores(rng, n, m):
    out = []
    for _ in range(m):
        D = rng.random((n, n)) * 3
        lab = rng.integers(0, max(2, n // 5), n)
        D[lab[:, None] == lab[None, :]] *= 0.3           # clique structure
        out.append(D)
    return out


@pytest.mark.parametrize("n,m,K,niters", [(30, 2, 5, 3), (61, 3, 20, 20), (200, 4, 20, 20), (257, 3, 7, 5), (700, 2, 64, 2)])
def test_fused_matrix_matches_oracle(ctx, n, m, K, niters):
    """acx_snf_fuse_dists (affinity matrices, neighbour lists and diffusion on the device) and
    acx_snf_fuse (lists from the caller) against the oracle's restatement of the reference."""
    import oracle
    from acoss_amd.algorithms import similarity_fusion as sf
    rng = np.random.default_rng(n + m)
    Scores = _scores(rng, n, m)
    Ws_o, want = oracle.snf_fuse(Scores, K=K, niters=niters, reg_diag=1)
    Ws_d, dev = sf.doSimilarityFusion(Scores, K=K, niters=niters, reg_diag=1, ctx=ctx)
    for a, b in zip(Ws_o, Ws_d):
        np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(dev, want, rtol=1e-10, atol=1e-14)
    assert sf.doSimilarityFusion(Scores, K=K, niters=1, reg_diag=1, ctx=ctx, want_ws=False)[0] is None
    if n == 30:     # the reference's own call, positional, no libacx vocabulary (similarity_fusion.py:188): default context
        Ws_b, bare = sf.doSimilarityFusion(Scores, K, niters, 1)
        assert np.array_equal(bare, dev) and all(np.array_equal(a, b) for a, b in zip(Ws_b, Ws_d))
    lists = [oracle.snf_knn_lists(W, K) for W in Ws_o]
    dev2 = ctx.snf_fuse(Ws_o, [l[0] for l in lists], [l[1] for l in lists], niters, 1.0)
    np.testing.assert_allclose(dev2, want, rtol=1e-10, atol=1e-14)
    # reg_diag = 0 and heavy ties in the neighbour ranking (identical rows): the cut is in column order
    Scores[0][:, :] = 1.0
    np.testing.assert_allclose(sf.doSimilarityFusion(Scores, K=K, niters=2, reg_diag=0, ctx=ctx)[1],
                               oracle.snf_fuse(Scores, K=K, niters=2, reg_diag=0)[1], rtol=1e-10, atol=1e-14)
    # non-symmetric input (EarlyFusion's 1 / (1 + D) of a mirrored matrix is symmetric, Chen's sqrt(T_j) / D is not)
    A = [rng.random((n, n)) * 3 for _ in range(m)]
    np.testing.assert_allclose(sf.doSimilarityFusion(A, K=K, niters=2, reg_diag=1, ctx=ctx)[1],
                               oracle.snf_fuse(A, K=K, niters=2, reg_diag=1)[1], rtol=1e-10, atol=1e-14)


def test_late_fusion_of_the_host_classes(ctx, tmp_path, monkeypatch):
    """EarlyFusion.do_late_fusion / ChenFusion.do_late_fusion (earlyfusion_traile.py:200-206,
    latefusion_chen.py:87-91) on score matrices filled by hand."""
    import oracle
    from acoss_amd.algorithms.earlyfusion_traile import EarlyFusion
    rng = np.random.default_rng(0)
    ef = EarlyFusion.__new__(EarlyFusion)
    ef._ctx, ef._device, ef.Ds = ctx, 0, {}
    for s in ("mfccs", "ssms", "chromas", "early"):
        D = (rng.random((40, 40)) * 5).astype(np.float32)
        ef.Ds[s] = D + D.T
    ef.do_late_fusion()
    inv = lambda s: 1.0 / (1.0 + np.array(ef.Ds[s], dtype=np.float64))
    np.testing.assert_allclose(ef.Ds["late"], oracle.snf_fuse([inv(s) for s in ("chromas", "ssms", "mfccs")], K=20, niters=20, reg_diag=1)[1],
                               rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(ef.Ds["early+late"],
                               oracle.snf_fuse([inv(s) for s in ("chromas", "ssms", "mfccs", "early")], K=20, niters=20, reg_diag=1)[1],
                               rtol=1e-9, atol=1e-12)


def test_errors(ctx):
    W = np.eye(8)
    J = np.zeros((8, 3), np.int32)
    V = np.ones((8, 3))
    with pytest.raises(ValueError):
        ctx.snf_fuse([W], [J], [V])                       # m = 1: the mean over the other matrices is undefined
    with pytest.raises(ValueError):
        ctx.snf_fuse([W, W], [J + 8, J], [V, V])          # neighbour index out of range
    with pytest.raises(ValueError):
        ctx.snf_fuse([W, W], [J, J], [V, V], niters=0)
