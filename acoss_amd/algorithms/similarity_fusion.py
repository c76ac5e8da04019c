"""
Similarity network fusion (Wang et al. 2012/2014) on N x N score matrices -- the late-fusion
post-step of EarlyFusion.do_late_fusion (earlyfusion_traile.py:200-206) and
LateFusionChen.do_late_fusion (latefusion_chen.py:87-91).  Drop-in for doSimilarityFusion of
acoss/algorithms/utils/similarity_fusion.py:188-196; the whole of it -- affinity matrices (getW,
:15-36), row-normalised kNN kernels (getS, :124-144, ties at the cut in column order), getP and the
cross-diffusion loop (doSimilarityFusionWs, :146-186, including the reference's aliasing of its two
work lists) -- runs in libacx's HIP kernels (acx_snf_fuse_dists).  There is no host implementation
in this package: the numpy restatement that pins the kernels to the reference lives in oracle/.
f64 throughout (the reference works in the dtype of its float32 memmaps: results agree to ~1e-7
relative).
"""
import numpy as np

__all__ = ["doSimilarityFusion"]


_DEFAULT_CTX = {}


def _default_context():
    """The context a bare doSimilarityFusion(Scores, K, niters, reg_diag) call -- the reference's signature,
    similarity_fusion.py:188 -- runs on: one per process and GPU (LOCAL_RANK's, else device 0), created on first use
    and kept, so that a caller written against acoss needs no libacx vocabulary.  No GPU / no libacx.so: the
    constructor raises (there is no host implementation)."""
    import os
    from .. import _lib
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    if dev not in _DEFAULT_CTX:
        _DEFAULT_CTX[dev] = _lib.Context(dev)
    return _DEFAULT_CTX[dev]


def doSimilarityFusion(Scores, K=5, niters=5, reg_diag=1, ctx=None, want_ws=True):
    """(affinity matrices, fused matrix) from a list of N x N distance matrices -- the reference's signature and
    return value.  Extra keywords: `ctx`, a libacx context to run on (default: the process's default context, see
    _default_context); want_ws=False skips copying the affinity matrices back (at N = 15 000 they are 1.8 GB each)
    and returns None in their place."""
    if ctx is None:
        ctx = _default_context()
    return ctx.snf_fuse_dists([np.asarray(D) for D in Scores], K=K, niters=niters, reg_diag=reg_diag, want_ws=want_ws)
