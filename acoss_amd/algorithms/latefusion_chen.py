"""
ChenFusion: Qmax + Dmax late fusion (Chen, Li & Xiao 2018).  Drop-in for
acoss/algorithms/latefusion_chen.py: same constructor, load_features(i), similarity(idxs),
normalize_by_length(), do_late_fusion(), similarity types "qmax" / "dmax" (+ "Late").
The reference builds ONE essentia cross recurrence plot per pair and aligns it twice
(latefusion_chen.py:58-72); here libacx does the same for the whole `idxs` array in one call
(acx_chenfusion_pairs: band kernels once, the bitmap DP twice).
"""
import numpy as np

from .. import _lib
from .rqa_serra09 import Serra09
from .algorithm_template import CoverAlgorithm
from .similarity_fusion import doSimilarityFusion

__all__ = ["ChenFusion"]


class ChenFusion(Serra09):
    def __init__(self, dataset_csv, datapath, chroma_type='hpcp', shortname='benchmark',
                 oti=True, kappa=0.095, tau=1, m=9, downsample_fac=40, device=None, engine=None, nonfinite="raise"):
        self.oti = oti
        self.kappa = kappa
        self.tau = tau
        self.m = m
        self.chroma_type = chroma_type
        self.downsample_fac = downsample_fac
        self.all_feats = {}
        self._device = device
        self._nonfinite = nonfinite
        self._engine = dict(engine or {})
        self._ctx = None
        self._pool_ready = False
        self._pooled_len = None
        self._bind_collective_device(self._device)      # before the first collective of this object
        CoverAlgorithm.__init__(self, dataset_csv=dataset_csv, name="LateFusionChen", datapath=datapath,
                                shortname=shortname, similarity_types=["qmax", "dmax"])

    def _grid(self):
        return self._context(), _lib.ALGO_CHENFUSION, self._params(), ["qmax", "dmax"]

    def similarity(self, idxs):
        idxs = np.asarray(idxs).reshape(-1, 2)
        if len(idxs) == 0:
            return
        sc = self._context().chenfusion_pairs(idxs.astype(np.int32), self._params())
        self.Ds["qmax"][idxs[:, 0], idxs[:, 1]] = sc[:, 0]
        self.Ds["dmax"][idxs[:, 0], idxs[:, 1]] = sc[:, 1]

    def normalize_by_length(self):
        """D[i, j] = sqrt(T_j) / D[i, j] (latefusion_chen.py:75-85): a DISTANCE, smaller =
        closer; unfilled cells (the diagonal) become +inf exactly as in the reference."""
        if not self.owns_result():
            return
        norm = np.sqrt(self._pooled_lengths().astype(np.float64))
        for key in self.Ds.keys():
            D = self.Ds[key]
            with np.errstate(divide="ignore"):
                for j0 in range(0, self.N, 2048):
                    j1 = min(self.N, j0 + 2048)
                    D[:, j0:j1] = (norm[None, j0:j1] / D[:, j0:j1]).astype(np.float32)

    def do_late_fusion(self):
        """SNF of the two distance matrices (latefusion_chen.py:87-91): Ds["Late"] = fused
        similarity; the two inputs are negated so that larger = closer everywhere.  Runs on the GPU
        (acx_snf_fuse_dists) of the rank that owns the matrices; the other ranks return at once (their
        matrices are empty: 20 N x N sweeps over inf / NaN for nothing)."""
        if not self.owns_result():
            self.Ds["Late"] = np.zeros((0, 0), np.float64)     # same keys on every rank (getEvalStatistics is collective)
            return
        if self._ctx is None:
            import os
            self._ctx = _lib.Context(self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0")), nonfinite=getattr(self, "_nonfinite", "raise"))
        DLate = doSimilarityFusion([self.Ds[s] for s in self.Ds], K=20, niters=20, reg_diag=1, ctx=self._ctx,
                                   want_ws=False)[1]
        for key in self.Ds:
            self.Ds[key] *= -1
        self.Ds["Late"] = DLate
