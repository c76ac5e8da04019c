"""
EarlyFusion (Tralie 2017: early MFCC / HPCP fusion).  Drop-in for the per-pair and fusion
surface of acoss/algorithms/earlyfusion_traile.py: same constructor, similarity(idxs) writing
Ds['mfccs'|'ssms'|'chromas'|'early'], do_late_fusion().  The per-pair chain (three
cross-similarity matrices, row-kappa binarisation, constrained Smith-Waterman x4, kernel
fusion; earlyfusion_traile.py:157-198) runs in libacx's HIP kernels.

Block-feature preparation (beat-synchronous MFCC / SSM / chroma blocks, :100-154) runs on the
host (numpy + scipy.ndimage), once per track: `block_features()` / `resize_block()` below.  The
reference resizes blocks with skimage.transform.resize(anti_aliasing=True, mode='constant'),
a library that is neither pinned by the reference (not in its setup.py) nor installed here, so
that step is a restatement of skimage's published algorithm (Gaussian pre-filter with
sigma = (scale - 1) / 2, then order-1 scipy.ndimage.zoom on the pixel grid, then clipping to the
input range) -- PARITY UNPINNED for it; everything around it follows the reference line by line.
A track file that already holds block features (keys mfccs (nb,650), ssms (nb,1225), chromas
(nb,480), chroma_med (12,)) is used as is.
"""
import os

import numpy as np

from .. import _lib
from .algorithm_template import CoverAlgorithm
from .similarity_fusion import doSimilarityFusion

__all__ = ["EarlyFusion", "resize_block", "block_features"]


def resize_block(X, i1, i2, frames_per_block):
    """earlyfusion_traile.py:214-247 (median_aggregate=False): frames [i1, i2) of X resampled
    to `frames_per_block` rows as skimage.transform.resize(x, (frames_per_block, d),
    anti_aliasing=True, mode='constant') does (order 1, cval 0, clip to the input range),
    restated with the scipy.ndimage primitives skimage itself calls; inf / nan -> 0."""
    import scipy.ndimage as ndi
    x = np.asarray(X)[i1:i2, :].astype(np.float64)
    n = x.shape[0]
    if n == 0:
        return np.zeros((frames_per_block, x.shape[1]))
    factor = n / float(frames_per_block)
    sigma = max(0.0, (factor - 1.0) / 2.0)
    filt = ndi.gaussian_filter(x, (sigma, 0.0), cval=0.0, mode="grid-constant") if sigma > 0 else x
    out = ndi.zoom(filt, (frames_per_block / float(n), 1.0), order=1, mode="grid-constant", cval=0.0,
                   grid_mode=True)
    if out.shape[0] != frames_per_block:       # rounding of n * zoom
        fixed = np.zeros((frames_per_block, x.shape[1]))
        m = min(frames_per_block, out.shape[0])
        fixed[:m] = out[:m]
        out = fixed
    lo, hi = min(float(x.min()), 0.0), max(float(x.max()), 0.0)
    if np.isfinite(lo) and np.isfinite(hi):
        out = np.clip(out, lo, hi)
    out[~np.isfinite(out)] = 0
    return out


def _ssm(X):
    """cross_recurrence.py:9-28 (get_ssm): Euclidean self-similarity matrix, zero diagonal."""
    sq = np.sum(X ** 2, 1)
    d2 = sq[:, None] + sq[None, :] - 2 * X.dot(X.T)
    d2[d2 < 0] = 0
    np.fill_diagonal(d2, 0)
    return np.sqrt(d2)


def block_features(feats, chroma_type="hpcp", blocksize=20, mfccs_per_block=50, chromas_per_block=40):
    """earlyfusion_traile.py:100-140: beat-synchronous blocks of one track.
    feats: the per-track dictionary of the feature store -- feats[chroma_type] (T, 12),
    feats['mfcc_htk'] (n_coeffs, T'), feats['madmom_features']['onsets'] (frame indices of the
    beats).  Returns mfccs (nb, mfccs_per_block * n_coeffs) f32 (z-normalised blocks), ssms
    (nb, mfccs_per_block (mfccs_per_block - 1) / 2) f32, chromas (nb, chromas_per_block * 12)
    f32, chroma_med (12,), nb = n_beats - blocksize."""
    chroma = np.asarray(feats[chroma_type])
    mfcc = np.array(feats["mfcc_htk"], dtype=np.float64).T
    mfcc[np.isnan(mfcc)] = 0
    onsets = np.asarray(feats["madmom_features"]["onsets"]).astype(np.int64)
    n_blocks = max(0, len(onsets) - blocksize)
    out = {"mfccs": np.zeros((n_blocks, mfccs_per_block * mfcc.shape[1]), dtype=np.float32)}
    pix = np.arange(mfccs_per_block)
    I, J = np.meshgrid(pix, pix)
    out["ssms"] = np.zeros((n_blocks, mfccs_per_block * (mfccs_per_block - 1) // 2), dtype=np.float32)
    for b in range(n_blocks):
        x = resize_block(mfcc, onsets[b], onsets[b + blocksize - 1], mfccs_per_block)
        x = x - np.mean(x, 0)[None, :]
        nrm = np.sqrt(np.sum(x ** 2, 1))[:, None]
        nrm[nrm == 0] = 1
        xn = x / nrm
        out["mfccs"][b, :] = xn.flatten()
        out["ssms"][b, :] = _ssm(xn)[I < J]
    out["chromas"] = np.zeros((n_blocks, chromas_per_block * chroma.shape[1]), dtype=np.float32)
    out["chroma_med"] = np.median(chroma, axis=0)
    for b in range(n_blocks):
        out["chromas"][b, :] = resize_block(chroma, onsets[b], onsets[b + blocksize], chromas_per_block).flatten()
    return out

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
        if all(k in feats for k in _KEYS):
            self.all_block_feats[i] = {k: np.asarray(feats[k]) for k in _KEYS}
        else:
            import time
            tic = time.time()
            self.all_block_feats[i] = block_features(feats, self.chroma_type, self.blocksize,
                                                     self.mfccs_per_block, self.chromas_per_block)
            if self.log_times:
                self.times['features'].append(time.time() - tic)
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
            self._ctx = _lib.Context(dev if dev is not None else int(os.environ.get("LOCAL_RANK", "0")))
        return self._ctx

    def do_late_fusion(self, host=False):
        """SNF of 1/(1+D) over the three / four score matrices (earlyfusion_traile.py:200-206).  The
        cross-diffusion loop runs on the GPU (acx_snf_fuse; no device -> the call fails);
        host=True asks explicitly for the dense numpy form of the same definitions (small N, checks)."""
        def inv(s):
            return 1.0 / (1.0 + np.array(self.Ds[s], dtype=np.float64))
        ctx = None if host else self._fusion_context()
        self.Ds["late"] = doSimilarityFusion([inv(s) for s in ("chromas", "ssms", "mfccs")],
                                             K=20, niters=20, reg_diag=1, ctx=ctx)[1]
        self.Ds["early+late"] = doSimilarityFusion([inv(s) for s in ("chromas", "ssms", "mfccs", "early")],
                                                   K=20, niters=20, reg_diag=1, ctx=ctx)[1]
