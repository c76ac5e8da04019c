"""
Serra09: cross recurrence quantification (Serra, Serra & Andrzejak 2009, NJP 11 093017).
Drop-in for acoss/algorithms/rqa_serra09.py: same constructor, load_features(i),
similarity(idxs), normalize_by_length().  The per-pair arithmetic that the reference
delegates to essentia (ChromaCrossSimilarity + CoverSongSimilarity, rqa_serra09.py:60-67)
runs in libacx's HIP kernels for ALL rows of `idxs` in one call.
"""
import numpy as np

from .. import _lib
from .algorithm_template import CoverAlgorithm

__all__ = ["Serra09", "pool_median"]


def pool_median(chroma, fac):
    """Median over consecutive blocks of `fac` frames (the last block may be shorter):
    what librosa.util.sync(chroma.T, arange(0, T, fac), aggregate=np.median).T yields at
    rqa_serra09.py:51.  (T0, d) -> (ceil(T0 / fac), d), dtype preserved."""
    chroma = np.asarray(chroma)
    T0 = chroma.shape[0]
    nfull = T0 // fac
    parts = []
    if nfull:
        parts.append(np.median(chroma[:nfull * fac].reshape(nfull, fac, -1), axis=1))
    if T0 > nfull * fac:
        parts.append(np.median(chroma[nfull * fac:], axis=0, keepdims=True))
    return np.concatenate(parts, axis=0).astype(chroma.dtype, copy=False)


class Serra09(CoverAlgorithm):
    """
    Attributes (as in the reference): chroma_type, downsample_fac, all_feats, oti, kappa,
    tau, m.  Extra keywords (all optional, behind the reference's):
      device     the GPU (default: LOCAL_RANK or 0)
      nonfinite  "raise" (default): NaN / Inf features fail the upload naming the track; "zero": replaced by 0
      engine     dict of the details of essentia's arithmetic that are only RECALLED, not pinned (essentia is not
                 installable here; include/acx.h acx_serra09_params, oracle/acx_oracle.c):
                   pct_mode    0 (default) linear-interpolated percentile, an exact-integer position k returns d_(k);
                               1 essentia's formula as recalled, d_(floor k) (ceil k - k) + d_(ceil k) (k - floor k),
                                 which is 0 at an exact-integer k (rows of 201, 401, ... cells at kappa = 0.095:
                                 the row binarises to nothing) -- pass engine={"pct_mode": 1} to reproduce that;
                               2 lower, 3 nearest
                   embed_full  0 (default) M = T - m tau frames, 1: T - (m - 1) tau
                   oti_target  0 (default) the reference track is transposed, 1: the query
                   dp_start    2 (default) or 3: first row / column of the alignment recursion
                   inclusive   1 (default) d <= eps, 0: d <
                   gamma_o, gamma_e  gap penalties (essentia's defaults 0.5 / 0.5)
                 and, not an essentia detail but a speed option of the device (include/acx.h ACX_ARITH_F16X2):
                   arith       "exact" (default: the f32 Gram the CPU oracle reproduces bit for bit) or "f16x2" (m = 9 only):
                               the frame Gram from two-term fp16 splits on the f16 matrix pipe -- as accurate against f64,
                               +14 ... 27 % pairs/s, but not the same bits: about one score in seven moves (by 0.5 - 4.5), MAP
                               stays within 1e-4 on the cover sets of tests/test_gpu_serra09.py
                 tests/test_essentia_pin.py finds the combination that reproduces essentia wherever it is installed.
    """
    n_chunks = 1      # the whole pair list goes to the GPU in one similarity() call

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
        CoverAlgorithm.__init__(self, dataset_csv=dataset_csv, name="Serra09", datapath=datapath,
                                shortname=shortname)

    # ------------------------------------------------------------------ features
    def load_features(self, i):
        if i not in self.all_feats:
            feats = CoverAlgorithm.load_features(self, i)
            self.all_feats[i] = pool_median(feats[self.chroma_type], self.downsample_fac)
        return self.all_feats[i]

    def set_pooled_features(self, tracks, labels=None):
        """Inject already-pooled (T_i, 12) chroma for every track (synthetic benchmarks,
        or features prepared elsewhere) instead of reading feature files."""
        assert len(tracks) == self.N
        self.all_feats = {i: np.ascontiguousarray(t, dtype=np.float32) for i, t in enumerate(tracks)}
        if labels is not None:
            for i, l in enumerate(labels):
                self._register_label(i, l)
        self._pool_ready = False

    # ------------------------------------------------------------------ device
    def _params(self):
        return _lib.serra09_params(m=self.m, tau=self.tau, kappa=self.kappa, oti=self.oti, **self._engine)

    def _context(self):
        if self._ctx is None:
            import os
            dev = self._device
            if dev is None:
                dev = int(os.environ.get("LOCAL_RANK", "0"))
            self._ctx = _lib.Context(dev, nonfinite=getattr(self, "_nonfinite", "raise"))
        if not self._pool_ready:
            if len(self.all_feats) == self.N or self.downsample_fac > 64:
                # pooled features injected (set_pooled_features) or already prepared by load_features
                tracks = [np.ascontiguousarray(self.load_features(i), dtype=np.float32) for i in range(self.N)]
                lens = np.array([t.shape[0] for t in tracks], dtype=np.int64)
                offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                self._ctx.upload_pool(np.concatenate(tracks, axis=0), offsets)
            else:
                # raw chroma of every track -> block medians on the device (acx_upload_raw_pool)
                raw = [np.ascontiguousarray(CoverAlgorithm.load_features(self, i)[self.chroma_type], dtype=np.float32)
                       for i in range(self.N)]
                offsets = np.concatenate([[0], np.cumsum([t.shape[0] for t in raw])]).astype(np.int64)
                lens = np.diff(self._ctx.upload_raw_pool(np.concatenate(raw, axis=0), offsets, self.downsample_fac))
            self._pooled_len = np.asarray(lens, dtype=np.int64)
            self._pool_ready = True
            self._warn_exact_percentile_positions()
        return self._ctx

    def exact_percentile_tracks(self):
        """Tracks whose embedded length M puts the kappa-percentile of a row / column of M cells on an EXACT-INTEGER
        position k = (M - 1) kappa (f32, as the kernels compute it): with kappa = 0.095f that is M - 1 = 200, 400, 600,
        i.e. pooled lengths 210 / 410 / 610.  There -- and only there -- the two recalled forms of essentia's percentile
        differ: engine pct_mode 0 (default) takes the order statistic d_(k), pct_mode 1 (the d0 + d1 form as recalled,
        include/acx.h) yields 0 and the row binarises to nothing.  Neither is evidenced while essentia is absent
        (DESIGN.md section 2): every pair such a track takes part in depends on the choice."""
        p = self._params()
        span = (int(self.m) - 1) * int(self.tau) if p.embed_full else int(self.m) * int(self.tau)
        M = (self._pooled_lengths().astype(np.int64) - span + int(self.tau) - 1) // int(self.tau)   # acx_serra09_embed_len
        k = (np.maximum(M, 2) - 1).astype(np.float32) * np.float32(self.kappa)
        return np.nonzero((M > 1) & (k == np.floor(k)))[0]

    def _warn_exact_percentile_positions(self):
        if "pct_mode" in self._engine:
            return                                      # the caller chose
        hit = self.exact_percentile_tracks()
        if len(hit):
            import warnings
            warnings.warn("Serra09: %d of %d tracks (first: %s) have an embedded length whose kappa-percentile position is an "
                          "exact integer, where the recalled forms of essentia's percentile differ (engine={'pct_mode': 0} "
                          "order statistic -- used -- vs {'pct_mode': 1} zero threshold); pass engine={'pct_mode': ...} to "
                          "choose explicitly" % (len(hit), self.N, hit[:8].tolist()), RuntimeWarning, stacklevel=3)

    def _pooled_lengths(self):
        if self._pool_ready:
            return self._pooled_len
        return np.array([self.load_features(j).shape[0] for j in range(self.N)], dtype=np.int64)

    def _grid(self):
        """all_pairwise runs the whole N x N grid inside libacx (algorithm_template._all_pairwise_grid)."""
        return self._context(), _lib.ALGO_SERRA09, self._params(), ["main"]

    def similarity(self, idxs):
        idxs = np.asarray(idxs).reshape(-1, 2)
        if len(idxs) == 0:
            return
        scores = self._context().serra09_pairs(idxs.astype(np.int32), self._params())
        for key in self.Ds.keys():
            self.Ds[key][idxs[:, 0], idxs[:, 1]] = scores

    def normalize_by_length(self):
        """Non-symmetric normalisation: D[i, j] /= sqrt(T_j), T_j the pooled length
        (rqa_serra09.py:71-83; the reciprocal of the paper's distance, so larger = closer)."""
        if not self.owns_result():
            return
        norm = np.sqrt(self._pooled_lengths().astype(np.float64))
        for key in self.Ds.keys():
            D = self.Ds[key]
            for j0 in range(0, self.N, 2048):
                j1 = min(self.N, j0 + 2048)
                D[:, j0:j1] = (D[:, j0:j1] / norm[None, j0:j1]).astype(np.float32)
