"""
GPU parity test for the similarity-network-fusion loop (acx_snf_fuse) through the C ABI: against the
host numpy restatement (acoss_amd.algorithms.similarity_fusion, dense) and the oracle's
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


@pytest.mark.parametrize("n,m,K,niters", [(30, 2, 5, 3), (61, 3, 20, 20), (200, 4, 20, 20), (257, 3, 7, 5)])
def test_fused_matrix_matches_host_and_oracle(ctx, n, m, K, niters):
    import oracle
    from acoss_amd.algorithms import similarity_fusion as sf
    rng = np.random.default_rng(n + m)
    Scores = _scores(rng, n, m)
    Ws_h, host = sf.doSimilarityFusion(Scores, K=K, niters=niters, reg_diag=1)
    Ws_d, dev = sf.doSimilarityFusion(Scores, K=K, niters=niters, reg_diag=1, ctx=ctx)
    assert all(np.array_equal(a, b) for a, b in zip(Ws_h, Ws_d))
    np.testing.assert_allclose(dev, host, rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(dev, oracle.snf_fuse(Scores, K=K, niters=niters, reg_diag=1)[1], rtol=1e-10, atol=1e-14)
    # reg_diag = 0 and heavy ties in the neighbour ranking (identical rows)
    Scores[0][:, :] = 1.0
    np.testing.assert_allclose(sf.doSimilarityFusion(Scores, K=K, niters=2, reg_diag=0, ctx=ctx)[1],
                               sf.doSimilarityFusion(Scores, K=K, niters=2, reg_diag=0)[1], rtol=1e-10, atol=1e-14)


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
