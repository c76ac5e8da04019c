"""
Serra09 sub-steps against the reference's OWN numpy code (tests/golden/serra09_substeps.npz, written by
tests/golden/make_serra09_substeps.py from /root/reference in the authoring container):

  * the optimal transposition index                == cross_recurrence.get_oti (cross_recurrence.py:75-103)
  * the distance matrix of the 108-dim stacked frames ~ cross_recurrence.get_csm (cross_recurrence.py:30-48)
  * the cross recurrence plot                      == get_csm's matrix -> numpy's linear percentile (kappa) of every
                                                      row and column -> mutual threshold (Serra et al. 2009 eq. 3)
  * normalize_by_length                            == Serra09.normalize_by_length (rqa_serra09.py:71-83), run through
                                                      the reference class with a stub essentia import

for the oracle (both arithmetics; CPU) and for the device through the C ABI (`-m gpu`).  What stays unpinned
after this is the alignment recursion itself (essentia's CoverSongSimilarity) and the recalled switches of
SURVEY App. C; DESIGN.md section 2 has the table.

Tolerance on the distances (floating point, stated): |d^2 - d^2_f64| <= 1e-6 (|x|^2 + |y|^2) against the reference's
matrix evaluated in f64 -- the reference's own f32 evaluation sits at 5.5e-7 on the same scale --, and
|d - d_ref32| <= 2e-6 d_ref32 on the well-separated i.i.d. pair (cover pairs have near-identical frames whose
distances cancel, where f32 rounding of either side is 1e-4 relative).  Everything else is exact.
"""
import numpy as np
import pytest

N_TOL = 1e-6


def _z(golden):
    return golden("serra09_substeps")


def _embed_norms(x, M):
    X = np.concatenate([x[k:k + M] for k in range(9)], axis=1).astype(np.float64)
    return (X ** 2).sum(axis=1)


def _ref_plot(c, kappa=0.095, dtype=np.float32):
    """The reference's matrix -> thresholds by numpy's linear-interpolated percentile -> mutual recurrence plot."""
    er = np.percentile(c, 100 * kappa, axis=1, method="linear").astype(dtype)
    ec = np.percentile(c, 100 * kappa, axis=0, method="linear").astype(dtype)
    return ((c <= er[:, None]) & (c <= ec[None, :])).astype(np.uint8)


def _check_distances(z, k, d, oti, tag):
    q, r = z["p%d_q" % k], z["p%d_r" % k]
    M, N = d.shape
    assert oti == int(z["p%d_oti" % k]), "%s pair %d: OTI %d, reference get_oti %d" % (tag, k, oti, int(z["p%d_oti" % k]))
    c32, c64 = z["p%d_csm32" % k][:M, :N], z["p%d_csm64" % k][:M, :N]
    rr = np.roll(r, oti, axis=1)
    scale = _embed_norms(q, M)[:, None] + _embed_norms(rr, N)[None, :]
    err = np.abs(d.astype(np.float64) ** 2 - c64 ** 2) / scale
    assert err.max() <= N_TOL, "%s pair %d: d^2 off by %.3g (|x|^2 + |y|^2)" % (tag, k, err.max())
    if c64.min() > 1.0:                                     # no cancellation: plain relative error against the f32 reference
        rel = np.abs(d - c32) / c32
        assert rel.max() <= 2e-6, "%s pair %d: relative distance error %.3g" % (tag, k, rel.max())


@pytest.mark.parametrize("arith", ["seq108", "tree"])
def test_oracle_oti_and_stacked_csm_against_reference(golden, arith):
    import oracle
    z = _z(golden)
    for k in range(int(z["n_pairs"])):
        s, it = oracle.serra09_pair(z["p%d_q" % k], z["p%d_r" % k], oracle.serra09_params(arith=arith),
                                    want_intermediates=True)
        _check_distances(z, k, it["d"], it["oti"], "oracle[%s]" % arith)
        # the other OTI convention is get_oti with the arguments swapped
        s1, it1 = oracle.serra09_pair(z["p%d_q" % k], z["p%d_r" % k], oracle.serra09_params(arith=arith, oti_target=1),
                                      want_intermediates=True)
        assert it1["oti"] == int(z["p%d_oti_query" % k])


@pytest.mark.parametrize("arith", ["seq108", "tree"])
def test_oracle_recurrence_plot_from_reference_matrix(golden, arith):
    """The plot the oracle binarises == the plot obtained from the REFERENCE's distance matrix (f32 and f64) with numpy's
    percentile, cell for cell; and the oracle's alignment of that plot gives the oracle's score."""
    import oracle
    z = _z(golden)
    for k in range(int(z["n_pairs"])):
        s, it = oracle.serra09_pair(z["p%d_q" % k], z["p%d_r" % k], oracle.serra09_params(arith=arith),
                                    want_intermediates=True)
        M, N = it["d"].shape
        for name, dt in (("csm32", np.float32), ("csm64", np.float64)):
            R = _ref_plot(z["p%d_%s" % (k, name)][:M, :N], dtype=dt)
            assert np.array_equal(R, it["R"]), "pair %d %s: %d cells differ" % (k, name, int(np.sum(R != it["R"])))
            assert oracle.qmax_binary(R) == s


def test_global_chroma_and_oti_tie(golden):
    import ctypes
    import oracle
    z = _z(golden)
    L = oracle.lib()
    fp = ctypes.POINTER(ctypes.c_float)
    for k in range(int(z["n_pairs"])):
        q = np.ascontiguousarray(z["p%d_q" % k], dtype=np.float32)
        g = np.zeros(12, np.float32)
        L.acx_o_global_chroma(q.ctypes.data_as(fp), q.shape[0], g.ctypes.data_as(fp))
        assert np.allclose(g, z["p%d_gq" % k], rtol=2e-6, atol=0)
    ones = np.ones(12, np.float32)
    assert L.acx_o_oti(ones.ctypes.data_as(fp), ones.ctypes.data_as(fp)) == int(z["oti_tie"]) == 0     # first maximum wins


def test_normalize_by_length_against_reference_class(golden):
    """rqa_serra09.py:71-83 run through the reference class: the oracle's restatement and the product's vectorised
    host code (no GPU involved) reproduce it bit for bit."""
    import oracle
    z = _z(golden)
    lengths, D_in, D_out = z["norm_lengths"], z["norm_D_in"], z["norm_D_out"]
    assert np.array_equal(oracle.serra09_normalize_by_length(D_in, lengths), D_out)
    # the product's method on an object that holds only what the method reads
    from acoss_amd.algorithms.rqa_serra09 import Serra09

    class Bare(Serra09):
        def __init__(self):
            self.N = len(lengths)
            self.Ds = {"main": D_in.copy()}
            self._pool_ready = False
            self.all_feats = {j: np.zeros((int(lengths[j]), 12), np.float32) for j in range(self.N)}

        def owns_result(self):
            return True
    b = Bare()
    b.normalize_by_length()
    assert b.Ds["main"].dtype == np.float32
    assert np.array_equal(b.Ds["main"], D_out)


@pytest.mark.gpu
def test_device_oti_distances_plot_against_reference(golden):
    """The device's OTI, squared distances and recurrence plot (acx_serra09_debug_pair, through the C ABI) against the
    reference's get_oti / get_csm goldens; and its score == the oracle's alignment of the reference-derived plot."""
    import oracle
    from acoss_amd import _lib
    z = _z(golden)
    n = int(z["n_pairs"])
    tracks = []
    for k in range(n):
        tracks += [z["p%d_q" % k], z["p%d_r" % k]]
    offs = np.concatenate([[0], np.cumsum([t.shape[0] for t in tracks])]).astype(np.int64)
    ctx = _lib.Context(0)
    try:
        ctx.upload_pool(np.concatenate(tracks), offs)
        batch = ctx.serra09_pairs(np.array([[2 * k, 2 * k + 1] for k in range(n)], np.int32))
        for k in range(n):
            g = ctx.serra09_debug_pair(2 * k, 2 * k + 1)
            d = np.sqrt(g["d2"])
            _check_distances(z, k, d, g["oti"], "device")
            M, N = d.shape
            Rg = ((g["d2"] <= g["thr_q"][:, None]) & (g["d2"] <= g["thr_r"][None, :])).astype(np.uint8)
            for name, dt in (("csm32", np.float32), ("csm64", np.float64)):
                R = _ref_plot(z["p%d_%s" % (k, name)][:M, :N], dtype=dt)
                assert np.array_equal(R, Rg), "pair %d %s: %d cells differ" % (k, name, int(np.sum(R != Rg)))
            assert oracle.qmax_binary(Rg) == g["score"] == float(batch[k])
            assert ctx.qmax_binary(Rg) == g["score"]
    finally:
        ctx.close()


def test_exact_percentile_positions_are_reported():
    """Pooled lengths 210 / 410 / 610 (embedded rows of 201 / 401 / 601 cells at kappa = 0.095f) are where the recalled
    forms of essentia's percentile differ (oracle pct_mode 0 vs 1): the class names those tracks and warns unless the
    caller chose a pct_mode.  The oracle confirms the position arithmetic: only there do modes 0 and 1 disagree."""
    import warnings
    import oracle
    from acoss_amd.algorithms.rqa_serra09 import Serra09
    lengths = np.array([150, 209, 210, 211, 410, 500, 610, 650])

    class Bare(Serra09):
        def __init__(self, engine=None):
            self.N, self.m, self.tau, self.kappa, self.oti = len(lengths), 9, 1, 0.095, True
            self._engine = dict(engine or {})
            self._pool_ready = True
            self._pooled_len = lengths
    b = Bare()
    assert b.exact_percentile_tracks().tolist() == [2, 4, 6]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        b._warn_exact_percentile_positions()
        Bare(engine={"pct_mode": 0})._warn_exact_percentile_positions()
    assert len(w) == 1 and "pct_mode" in str(w[0].message)
    rng = np.random.default_rng(3)
    q = rng.random((120, 12), dtype=np.float32)
    for T, differs in ((209, False), (210, True), (211, False)):
        r = rng.random((T, 12), dtype=np.float32)
        e0 = oracle.serra09_pair(q, r, oracle.serra09_params(pct_mode=0), want_intermediates=True)[1]["eps_q"]
        e1 = oracle.serra09_pair(q, r, oracle.serra09_params(pct_mode=1), want_intermediates=True)[1]["eps_q"]
        assert (not np.array_equal(e0, e1)) == differs
        if differs:
            assert np.all(e1 == 0)


class _BareChen(object):
    """ChenFusion holding only what normalize_by_length / do_late_fusion read (no dataset files)."""

    @staticmethod
    def make(z, device=None):
        from acoss_amd.algorithms.latefusion_chen import ChenFusion

        class Bare(ChenFusion):
            def __init__(self):
                lengths = z["chen_lengths"]
                self.N = len(lengths)
                self.Ds = {"qmax": z["chen_q_in"].copy(), "dmax": z["chen_d_in"].copy()}
                self._pool_ready = False
                self._ctx = None
                self._device = device
                self._nonfinite = "raise"
                self.all_feats = {j: np.zeros((int(lengths[j]), 12), np.float32) for j in range(self.N)}

            def owns_result(self):
                return True
        return Bare()


def test_chenfusion_normalisation_against_reference_class(golden):
    """latefusion_chen.py:75-85 run through the reference class (D = sqrt(T_j) / D, the unfilled diagonal -> inf): the
    product's vectorised host code reproduces it bit for bit; the oracle's restatement of the fusion that follows
    (latefusion_chen.py:87-91) reproduces the reference's fused matrix."""
    import oracle
    z = _z(golden)
    b = _BareChen.make(z)
    with np.errstate(divide="ignore"):
        b.normalize_by_length()
    assert np.array_equal(b.Ds["qmax"], z["chen_q_norm"]) and np.array_equal(b.Ds["dmax"], z["chen_d_norm"])
    assert np.isinf(np.diag(b.Ds["qmax"])).all()
    with np.errstate(all="ignore"):
        F = oracle.snf_fuse([z["chen_q_norm"], z["chen_d_norm"]], K=20, niters=20, reg_diag=1)[1]
    # the reference fuses in the dtype of its float32 memmaps, the oracle in f64: agreement to f32 rounding
    np.testing.assert_allclose(F, z["chen_late"], rtol=2e-5, atol=1e-7)
    assert np.array_equal(z["chen_q_after"], -z["chen_q_norm"])


@pytest.mark.gpu
def test_chenfusion_late_fusion_against_reference_class(golden):
    """ChenFusion.normalize_by_length + do_late_fusion of the product (SNF on the device) against the matrices the
    REFERENCE class produced from the same score matrices."""
    z = _z(golden)
    b = _BareChen.make(z, device=0)
    with np.errstate(divide="ignore"):
        b.normalize_by_length()
    b.do_late_fusion()
    try:
        np.testing.assert_allclose(np.asarray(b.Ds["Late"]), z["chen_late"], rtol=2e-5, atol=1e-7)
        assert np.array_equal(b.Ds["qmax"], z["chen_q_after"])
    finally:
        b._ctx.close()
