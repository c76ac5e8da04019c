"""
SiMPle: similarity matrix profile (Silva, Yeh, Batista & Keogh, ISMIR 2016).  Drop-in for
acoss/algorithms/simple_silva.py: same constructor and methods.  Feature preparation (mean
pooling, Hann smoothing, L2 column normalisation, simple_silva.py:34-43,56-66) runs ONCE
per track (the reference redoes it for every pair) -- on the device for the whole collection
(acx_simple_upload_raw_pool) when the tracks come from feature files, on the host in
load_features(i) for single-track access; OTI + matrix profile + median of every ordered pair
run in libacx's HIP kernel, in f64 like the reference.
"""
import numpy as np

from .. import _lib
from .algorithm_template import CoverAlgorithm

__all__ = ["Simple"]


def _hann_sym(n):
    """scipy.signal.get_window('hann', n, fftbins=False)."""
    if n == 1:
        return np.ones(1)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / (n - 1))


class Simple(CoverAlgorithm):
    """
    SSLEN: subsequence length; WIN / SKIP: window and hop of the dimensionality reduction.
    """
    n_chunks = 1

    def __init__(self, dataset_csv, datapath, chroma_type='hpcp', shortname='Covers80',
                 SSLEN=10, WIN=200, SKIP=100, device=None, nonfinite="raise"):
        self.SSLEN = SSLEN
        self.WIN = WIN
        self.SKIP = SKIP
        self.chroma_type = chroma_type
        self.all_feats = {}
        self._device = device
        self._nonfinite = nonfinite
        self._ctx = None
        self._pool_ready = False
        self._bind_collective_device(self._device)      # before the first collective of this object
        CoverAlgorithm.__init__(self, dataset_csv=dataset_csv, name="SiMPle", datapath=datapath,
                                shortname=shortname)

    # ------------------------------------------------------------------ features (host)
    def load_features(self, i, do_plot=False):
        if i not in self.all_feats:
            feats = CoverAlgorithm.load_features(self, i)
            feat_orig = np.asarray(feats[self.chroma_type]).T
            n = int(feat_orig.shape[1] / self.SKIP)
            pooled = np.zeros((feat_orig.shape[0], n))
            for k in range(n):
                pooled[:, k] = np.mean(feat_orig[:, k * self.SKIP:k * self.SKIP + self.WIN], axis=1)
            self.all_feats[i] = self.smooth(pooled)
        return self.all_feats[i]

    def set_features(self, feats, labels=None):
        """Inject ready (12, n_i) f64 features for every track (synthetic benchmarks)."""
        assert len(feats) == self.N
        self.all_feats = {i: np.asarray(f, dtype=np.float64) for i, f in enumerate(feats)}
        if labels is not None:
            for i, l in enumerate(labels):
                self._register_label(i, l)
        self._pool_ready = False

    def smooth(self, feat, win_len_smooth=4):
        """Hann(win+2, symmetric) / sum along time ('same', zero fill), then unit L2 columns
        (librosa.util.normalize: columns with a norm below tiny stay unscaled)."""
        n = win_len_smooth + 2
        win = _hann_sym(n)
        win = win / np.sum(win)
        feat = np.asarray(feat, dtype=np.float64)
        out = np.empty_like(feat)
        lo = (n - 1) // 2
        for c in range(feat.shape[0]):
            out[c] = np.convolve(feat[c], win, mode="full")[lo:lo + feat.shape[1]]
        nrm = np.sqrt(np.sum(out ** 2, axis=0, keepdims=True))
        nrm[nrm < np.finfo(np.float64).tiny] = 1.0
        return out / nrm

    def oti(self, seq_a, seq_b):
        """(seq_b rolled to best match seq_a, ascending argsort of the 12 shift scores)."""
        pa, pb = np.sum(seq_a, 1), np.sum(seq_b, 1)
        v = np.array([np.dot(pa, np.roll(pb, s, axis=0)) for s in range(12)])
        order = np.argsort(v)
        return np.roll(seq_b, order[-1], axis=0), order

    # ------------------------------------------------------------------ device
    def _context(self):
        if self._ctx is None:
            import os
            dev = self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0"))
            self._ctx = _lib.Context(dev, nonfinite=getattr(self, "_nonfinite", "raise"))
        if not self._pool_ready:
            if len(self.all_feats) == self.N:
                # features injected (set_features) or already prepared by load_features
                tracks = [np.ascontiguousarray(self.load_features(i).T, dtype=np.float64) for i in range(self.N)]
                offs = np.concatenate([[0], np.cumsum([t.shape[0] for t in tracks])]).astype(np.int64)
                self._ctx.upload_pool_f64(np.concatenate(tracks, axis=0), offs)
            else:
                # raw chroma of every track -> pooling, smoothing and normalisation on the device
                raw = [np.ascontiguousarray(CoverAlgorithm.load_features(self, i)[self.chroma_type], dtype=np.float32)
                       for i in range(self.N)]
                offs = np.concatenate([[0], np.cumsum([t.shape[0] for t in raw])]).astype(np.int64)
                self._ctx.simple_upload_raw_pool(np.concatenate(raw, axis=0), offs, self.WIN, self.SKIP, 4)
            self._pool_ready = True
        return self._ctx

    def simple_sim(self, seq_a, seq_b):
        """median_i min_j ||A[:, i:i+L] - B[:, j:j+L]||^2 of two (12, n) sequences (no OTI).  The two
        sequences become a two-track pool of a side context that lives as long as this object (the
        collection's pool in the main context stays where it is)."""
        import os
        if getattr(self, "_aux_ctx", None) is None:
            self._aux_ctx = _lib.Context(self._device if self._device is not None else int(os.environ.get("LOCAL_RANK", "0")),
                                         nonfinite=getattr(self, "_nonfinite", "raise"))
        ctx = self._aux_ctx
        a = np.ascontiguousarray(np.asarray(seq_a, dtype=np.float64).T)
        b = np.ascontiguousarray(np.asarray(seq_b, dtype=np.float64).T)
        ctx.upload_pool_f64(np.concatenate([a, b]), np.array([0, len(a), len(a) + len(b)], np.int64))
        return float(-ctx.simple_pairs(np.array([[0, 1]], np.int32), self.SSLEN, oti=False)[0])

    def _grid(self):
        return self._context(), _lib.ALGO_SIMPLE, _lib.SimpleParams(int(self.SSLEN), 1), ["main"]

    def similarity(self, idxs):
        idxs = np.asarray(idxs).reshape(-1, 2)
        if len(idxs) == 0:
            return
        sim = self._context().simple_pairs(idxs.astype(np.int32), self.SSLEN)
        self.Ds['main'][idxs[:, 0], idxs[:, 1]] = sim
