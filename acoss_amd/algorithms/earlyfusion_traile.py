"""
EarlyFusion (Tralie 2017: early MFCC / HPCP fusion).  Drop-in for the per-pair and fusion
surface of acoss/algorithms/earlyfusion_traile.py: same constructor, similarity(idxs) writing
Ds['mfccs'|'ssms'|'chromas'|'early'], do_late_fusion().  The per-pair chain (three
cross-similarity matrices, row-kappa binarisation, constrained Smith-Waterman x4, kernel
fusion; earlyfusion_traile.py:157-198) runs in libacx's HIP kernels.

Block-feature preparation (beat-synchronous MFCC / SSM / chroma blocks, :100-154, which needs
skimage.transform.resize) is outside this engine's scope for now: the class
reads ready block features -- keys mfccs (nb,650), ssms (nb,1225), chromas (nb,480),
chroma_med (12,) -- from the per-track feature file or from its cache / set_block_features().
"""
import os

import numpy as np

from .. import _lib
from .algorithm_template import CoverAlgorithm
from .similarity_fusion import doSimilarityFusion

__all__ = ["EarlyFusion"]

_KEYS = ("mfccs", "ssms", "chromas", "chroma_med")


class EarlyFusion(CoverAlgorithm):
    n_chunks = 1

    def __init__(self, dataset_csv, datapath, chroma_type='hpcp', shortname='Covers80', blocksize=20,
                 mfccs_per_block=50, ssm_res=50, chromas_per_block=40, kappa=0.1, K=10, niters=5,
                 log_times=False, device=None):
        self.chroma_type = chroma_type
        self.blocksize = blocksize
        self.mfccs_per_block = mfccs_per_block
        self.chromas_per_block = chromas_per_block
        self.kappa = kappa
        self.K = K
        self.niters = niters
        self.all_block_feats = {}
        self.log_times = log_times
        if log_times:
            self.times = {'features': [], 'raw': []}
        self._device = device
        self._ctx = None
        self._pool_ready = False
        CoverAlgorithm.__init__(self, dataset_csv=dataset_csv, name="EarlyFusionTraile", datapath=datapath,
                                shortname=shortname, similarity_types=["mfccs", "ssms", "chromas", "early"])

    def get_cacheprefix(self):
        return "%s/%s_%s_%s" % (self.cachedir, self.name, self.shortname, self.chroma_type)

    def load_features(self, i, do_plot=False):
        if i in self.all_block_feats:
            return self.all_block_feats[i]
        feats = CoverAlgorithm.load_features(self, i)
        if not all(k in feats for k in _KEYS):
            raise NotImplementedError(
                "EarlyFusion.load_features: %s holds no block features (%s). Building them from raw "
                "MFCC / chroma / beat onsets (earlyfusion_traile.py:100-154) is not part of the MI355X "
                "engine yet; store the block features in the track file or call set_block_features()."
                % (self.filepaths[i], ", ".join(_KEYS)))
        self.all_block_feats[i] = {k: np.asarray(feats[k]) for k in _KEYS}
        return self.all_block_feats[i]

    def set_block_features(self, tracks, labels=None):
        assert len(tracks) == self.N
        self.all_block_feats = dict(enumerate(tracks))
        if labels is not None:
            for i, l in enumerate(labels):
                self._register_label(i, l)
        self._pool_ready = False

    def _context(self):
        if self._ctx is None:
            dev = self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0"))
            self._ctx = _lib.Context(dev)
        if not self._pool_ready:
            self._ctx.ef_upload_pool([self.load_features(i) for i in range(self.N)])
            self._pool_ready = True
        return self._ctx

    def similarity(self, idxs, do_plot=False):
        idxs = np.asarray(idxs).reshape(-1, 2)
        if len(idxs) == 0:
            return
        sc = self._context().earlyfusion_pairs(idxs.astype(np.int32), kappa=self.kappa, K=self.K)
        for c, s in enumerate(("mfccs", "ssms", "chromas", "early")):
            self.Ds[s][idxs[:, 0], idxs[:, 1]] = sc[:, c]

    def do_late_fusion(self):
        """SNF of 1/(1+D) over the three / four score matrices (earlyfusion_traile.py:200-206)."""
        def inv(s):
            return 1.0 / (1.0 + np.array(self.Ds[s], dtype=np.float64))
        self.Ds["late"] = doSimilarityFusion([inv(s) for s in ("chromas", "ssms", "mfccs")],
                                             K=20, niters=20, reg_diag=1)[1]
        self.Ds["early+late"] = doSimilarityFusion([inv(s) for s in ("chromas", "ssms", "mfccs", "early")],
                                                   K=20, niters=20, reg_diag=1)[1]
