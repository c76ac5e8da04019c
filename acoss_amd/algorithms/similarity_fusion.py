"""
Similarity network fusion (Wang et al. 2012/2014) on N x N score matrices -- the late-fusion
post-step of EarlyFusion.do_late_fusion (earlyfusion_traile.py:200-206).  The affinity matrices
and neighbour lists are prepared in numpy on the host; the cross-diffusion loop runs on the GPU
(libacx acx_snf_fuse) when a context is passed, else in dense numpy; same definitions as acoss/algorithms/utils/similarity_fusion.py
(getW :15-36, getP :101-122, getS :124-144, doSimilarityFusionWs :146-186), with the kNN
truncation applied by a stable sort so that ties have one defined order.
"""
import numpy as np

__all__ = ["getW", "doSimilarityFusionWs", "doSimilarityFusion"]


def getW(D, K, Mu=0.5):
    """Affinity matrix from a (dis)similarity matrix: symmetrise, zero diagonal, local scale
    from the K nearest neighbours (mean of the K+1 smallest incl. the zero diagonal, rescaled)."""
    Dsym = 0.5 * (D + D.T)
    np.fill_diagonal(Dsym, 0)
    near = np.partition(Dsym, K + 1, 1)[:, 0:K + 1]
    mean_dist = np.mean(near, 1) * float(K + 1) / float(K)
    eps = (mean_dist[:, None] + mean_dist[None, :] + Dsym) / 3
    denom = 2 * (Mu * eps) ** 2
    denom[denom == 0] = 1
    return np.exp(-Dsym ** 2 / denom)


def _row_normalise(W):
    rs = np.sum(W, 1)
    rs[rs == 0] = 1
    return W / rs[:, None]


def _knn_lists(W, K):
    """The row-normalised K-nearest-neighbour kernel of W (getS) as K (column, weight) pairs per
    row: the K largest of each row, ties in column order (stable sort)."""
    n = W.shape[1]
    if K >= n:
        J = np.argsort(-W, 1, kind="stable")[:, :K]
    else:
        # == np.argsort(-W, 1, kind="stable")[:, :K] without sorting whole rows: argpartition finds the
        # K largest; only rows where the K-th value occurs more often than it was taken (ties across
        # the cut) are redone with the stable sort
        J = np.argpartition(-W, K - 1, 1)[:, :K]
        Vsel = np.take_along_axis(W, J, 1)
        thr = Vsel.min(1, keepdims=True)
        tied = np.nonzero(np.sum(W == thr, 1) > np.sum(Vsel == thr, 1))[0]
        order = np.lexsort((J, -Vsel), axis=1)                            # by value descending, then column
        J = np.take_along_axis(J, order, 1)
        if len(tied):
            J[tied] = np.argsort(-W[tied], 1, kind="stable")[:, :K]
    V = np.take_along_axis(W, J, 1)
    sn = np.sum(V, 1)
    sn[sn == 0] = 1
    return J.astype(np.int32), V / sn[:, None]


def _knn_kernel(W, K):
    n = W.shape[0]
    J, V = _knn_lists(W, K)
    S = np.zeros((n, n))
    np.put_along_axis(S, J.astype(np.int64), V, 1)
    return S


def doSimilarityFusionWs(Ws, K=5, niters=20, reg_diag=1, ctx=None):
    """Cross-diffusion of the affinity matrices; returns the fused N x N matrix.  Like the
    reference, from the second iteration on a matrix updated earlier in the same sweep is
    already seen by the later ones (the two work lists alias, similarity_fusion.py:179).
    With `ctx` (a libacx context) the loop runs on the GPU (acx_snf_fuse): same definitions, f64;
    without it in dense numpy on the host (O(N^3) per update: small N only)."""
    if ctx is not None:
        lists = [_knn_lists(np.asarray(W, dtype=np.float64), K) for W in Ws]
        return ctx.snf_fuse(Ws, [l[0] for l in lists], [l[1] for l in lists], niters, reg_diag)
    Ps = [_row_normalise(W) for W in Ws]
    Ss = [_knn_kernel(W, K) for W in Ws]
    m = len(Ps)
    n = Ws[0].shape[0]
    idx = np.arange(n)
    cur = [np.array(P) for P in Ps]
    nxt = [None] * m
    for it in range(niters):
        src = cur if it == 0 else nxt
        for i in range(m):
            acc = np.zeros((n, n))
            for k in range(m):
                if k != i:
                    acc += src[k]
            acc /= float(m - 1)
            upd = Ss[i].dot((Ss[i].dot(acc.T)).T)
            if reg_diag > 0:
                upd[idx, idx] += reg_diag
            nxt[i] = upd
    out = np.zeros((n, n))
    for P in nxt:
        out += P
    return out / m


def doSimilarityFusion(Scores, K=5, niters=5, reg_diag=1, ctx=None):
    """(affinity matrices, fused matrix) from a list of N x N distance matrices."""
    Ws = [getW(np.array(D, dtype=np.float64), K) for D in Scores]
    return Ws, doSimilarityFusionWs(Ws, K, niters, reg_diag, ctx=ctx)
