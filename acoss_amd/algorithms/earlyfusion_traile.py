"""
EarlyFusion (Tralie 2017: early MFCC / HPCP fusion).  Drop-in for the per-pair and fusion
surface of acoss/algorithms/earlyfusion_traile.py: same constructor, load_features(i),
similarity(idxs) writing Ds['mfccs'|'ssms'|'chromas'|'early'], do_late_fusion().

Everything numeric runs in libacx's HIP kernels:
  * block features (beat-synchronous MFCC / SSM / chroma blocks, earlyfusion_traile.py:100-140 with
    resize_block :214-247): acx_ef_block_features for one track (load_features(i)),
    acx_ef_upload_raw_pool for the whole collection, which never leaves the device.  The
    reference resizes blocks with skimage.transform.resize(anti_aliasing=True, mode='constant'), a
    library that is neither pinned by the reference (not in its setup.py) nor installed here: the
    kernel restates skimage's published algorithm -- PARITY UNPINNED for that step;
  * the per-pair chain (three cross-similarity matrices, row-kappa binarisation, constrained
    Smith-Waterman x4, kernel fusion; :157-198);
  * late fusion (similarity network fusion of the N x N score matrices, :200-206).
A track file that already holds block features (keys mfccs (nb,650), ssms (nb,1225), chromas
(nb,480), chroma_med (12,)) is used as is.
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
                 log_times=False, device=None, nonfinite="raise", engine=None):
        """The reference's arguments (earlyfusion_traile.py:40-56), plus: `device`, `nonfinite` ("raise" | "propagate" for
        NaN / inf in the features), and `engine`, a dict of the library's arithmetic switches for this chain --
        {"gemm": "f16x2" (default) | "bf16x3" | "f32" | ...} (Context.set_ef_gemm: how the three cross-similarity
        products are evaluated; all within an f32 sgemm's accuracy) and {"fuse": "fast" (default) | "exact"}
        (Context.set_ef_fuse: getWCSM's kernel weights by reciprocal + exp2, or in numpy's own operation order)."""
        self._engine = dict(engine or {})
        unknown = set(self._engine) - {"gemm", "fuse"}
        if unknown:
            raise ValueError("EarlyFusion: unknown engine option(s) %s" % sorted(unknown))
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
        self._nonfinite = nonfinite
        self._ctx = None
        self._pool_ready = False
        self._bind_collective_device(self._device)      # before the first collective of this object
        CoverAlgorithm.__init__(self, dataset_csv=dataset_csv, name="EarlyFusionTraile", datapath=datapath,
                                shortname=shortname, similarity_types=["mfccs", "ssms", "chromas", "early"])

    def get_cacheprefix(self):
        return "%s/%s_%s_%s" % (self.cachedir, self.name, self.shortname, self.chroma_type)

    def load_features(self, i, do_plot=False):
        if i in self.all_block_feats:
            return self.all_block_feats[i]
        feats = CoverAlgorithm.load_features(self, i)
        if all(k in feats for k in _KEYS):
            self.all_block_feats[i] = {k: np.asarray(feats[k]) for k in _KEYS}
        else:
            import time
            tic = time.time()
            raw = self._raw_track(feats)
            self.all_block_feats[i] = self._fusion_context().ef_block_features(
                raw["chroma"], raw["mfcc"], raw["onsets"], self.blocksize, self.mfccs_per_block, self.chromas_per_block)
            if self.log_times:
                self.times['features'].append(time.time() - tic)
        return self.all_block_feats[i]

    def _raw_track(self, feats):
        """What the block features are made of (earlyfusion_traile.py:103-110): the chroma (T, 12), the
        MFCCs time-major (feats['mfcc_htk'].T) and the beat onsets in frames."""
        return dict(chroma=np.asarray(feats[self.chroma_type], dtype=np.float32),
                    mfcc=np.ascontiguousarray(np.asarray(feats["mfcc_htk"], dtype=np.float32).T),
                    onsets=np.asarray(feats["madmom_features"]["onsets"]).astype(np.int64))

    def set_block_features(self, tracks, labels=None):
        assert len(tracks) == self.N
        self.all_block_feats = dict(enumerate(tracks))
        if labels is not None:
            for i, l in enumerate(labels):
                self._register_label(i, l)
        self._pool_ready = False

    def _context(self):
        self._fusion_context()
        if not self._pool_ready:
            feats = None
            if not self.all_block_feats:
                feats = [CoverAlgorithm.load_features(self, i) for i in range(self.N)]
            if feats is not None and not any(all(k in f for k in _KEYS) for f in feats):
                # raw features of every track -> block features built and kept on the device
                self._ctx.ef_upload_raw_pool([self._raw_track(f) for f in feats], self.blocksize,
                                             self.mfccs_per_block, self.chromas_per_block)
            else:
                self._ctx.ef_upload_pool([self.load_features(i) for i in range(self.N)])
            self._pool_ready = True
        return self._ctx

    def _grid(self):
        return (self._context(), _lib.ALGO_EARLYFUSION, _lib.EfParams(float(self.kappa), int(self.K)),
                ["mfccs", "ssms", "chromas", "early"])

    def similarity(self, idxs, do_plot=False):
        idxs = np.asarray(idxs).reshape(-1, 2)
        if len(idxs) == 0:
            return
        sc = self._context().earlyfusion_pairs(idxs.astype(np.int32), kappa=self.kappa, K=self.K)
        for c, s in enumerate(("mfccs", "ssms", "chromas", "early")):
            self.Ds[s][idxs[:, 0], idxs[:, 1]] = sc[:, c]

    def _fusion_context(self):
        """The GPU context of the pair grid, or a fresh one (scores loaded with precomputed=True)."""
        if getattr(self, "_ctx", None) is None:
            dev = getattr(self, "_device", None)
            self._ctx = _lib.Context(dev if dev is not None else int(os.environ.get("LOCAL_RANK", "0")), nonfinite=getattr(self, "_nonfinite", "raise"))
            eng = getattr(self, "_engine", {})
            if "gemm" in eng:
                self._ctx.set_ef_gemm(eng["gemm"])
            if "fuse" in eng:
                self._ctx.set_ef_fuse(eng["fuse"])
        return self._ctx

    def do_late_fusion(self):
        """SNF of 1/(1+D) over the three / four score matrices (earlyfusion_traile.py:200-206), on the
        GPU (acx_snf_fuse_dists; no device -> the call fails) of the rank that owns the matrices."""
        if not self.owns_result():
            for key in ("late", "early+late"):                  # same keys on every rank (getEvalStatistics is collective)
                self.Ds[key] = np.zeros((0, 0), np.float64)
            return

        def inv(s):
            return 1.0 / (1.0 + np.array(self.Ds[s], dtype=np.float64))
        ctx = self._fusion_context()
        self.Ds["late"] = doSimilarityFusion([inv(s) for s in ("chromas", "ssms", "mfccs")],
                                             K=20, niters=20, reg_diag=1, ctx=ctx, want_ws=False)[1]
        self.Ds["early+late"] = doSimilarityFusion([inv(s) for s in ("chromas", "ssms", "mfccs", "early")],
                                                   K=20, niters=20, reg_diag=1, ctx=ctx, want_ws=False)[1]
