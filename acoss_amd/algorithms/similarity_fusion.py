"""
Similarity network fusion (Wang et al. 2012/2014) on N x N score matrices -- the late-fusion
post-step of EarlyFusion.do_late_fusion (earlyfusion_traile.py:200-206).  Dense numpy host
code (O(N^2 K) per iteration); same definitions as acoss/algorithms/utils/similarity_fusion.py
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


def _knn_kernel(W, K):
    n = W.shape[0]
    J = np.argsort(-W, 1, kind="stable")[:, :K]
    V = np.take_along_axis(W, J, 1)
    sn = np.sum(V, 1)
    sn[sn == 0] = 1
    S = np.zeros((n, n))
    np.put_along_axis(S, J, V / sn[:, None], 1)
    return S


def doSimilarityFusionWs(Ws, K=5, niters=20, reg_diag=1):
    """Cross-diffusion of the affinity matrices; returns the fused N x N matrix.  Like the
    reference, from the second iteration on a matrix updated earlier in the same sweep is
    already seen by the later ones (the two work lists alias, similarity_fusion.py:179)."""
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


def doSimilarityFusion(Scores, K=5, niters=5, reg_diag=1):
    """(affinity matrices, fused matrix) from a list of N x N distance matrices."""
    Ws = [getW(np.array(D, dtype=np.float64), K) for D in Scores]
    return Ws, doSimilarityFusionWs(Ws, K, niters, reg_diag)
