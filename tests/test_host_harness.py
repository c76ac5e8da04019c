"""
CPU suite, part 4: the host-side mirror of the reference interface (CoverAlgorithm,
pair grid, symmetrisation, evaluation statistics, results CSV, feature files, Serra09
pooling / normalisation) against the goldens captured from the reference's own harness
and against the oracle.  No GPU needed: similarity() is overridden like a user subclass
(README.md:253-297 of the reference) would.
"""
import os

import numpy as np
import pytest

import oracle
from acoss_amd.algorithms.algorithm_template import CoverAlgorithm, eval_statistics
from acoss_amd.algorithms.rqa_serra09 import Serra09, pool_median
from acoss_amd.featurestore import load_track, save_track
from acoss_amd.utils import create_dataset_filepaths


def _toy_dataset(tmp_path, labels):
    csv = tmp_path / "toy.csv"
    with open(csv, "w") as f:
        f.write("work_id,track_id\n")
        for k, l in enumerate(labels):
            f.write("%s,t%d\n" % (l, k))
    root = str(tmp_path) + "/"
    for k, l in enumerate(labels):
        save_track(root + "%s/t%d.h5" % (l, k), {"label": l, "track_id": "t%d" % k,
                                                 "hpcp": np.zeros((3, 12), np.float32)})
    return str(csv), root


@pytest.mark.parametrize("sym", [True, False])
def test_harness_matches_reference_golden(golden, tmp_path, monkeypatch, sym):
    g = golden("harness")
    labels = [str(l) for l in g["labels"]]
    S = g["Strue"]
    csv, root = _toy_dataset(tmp_path, labels)
    monkeypatch.chdir(tmp_path)

    class Toy(CoverAlgorithm):
        def __init__(self):
            CoverAlgorithm.__init__(self, csv, name="Toy", datapath=root, shortname="toy")

        def similarity(self, idxs):
            for i, j in zip(idxs[:, 0], idxs[:, 1]):
                self.Ds["main"][i, j] = (0.5 * (S[i, j] + S[j, i])) if sym else S[i, j]

    toy = Toy()
    assert [p[len(root):] for p in toy.filepaths] == [str(p) for p in g["filepaths"]]
    assert isinstance(toy.Ds["main"], np.memmap) and toy.Ds["main"].shape == (14, 14)
    for k in range(toy.N):
        toy.load_features(k)
    toy.all_pairwise(parallel=0, symmetric=sym)
    tag = "sym" if sym else "asym"
    assert np.array_equal(np.array(toy.Ds["main"]), g["D_" + tag])
    res = toy.getEvalStatistics("main", topsidx=[1, 2, 5])
    got = np.array(list(res[:4]) + list(res[4]))
    np.testing.assert_allclose(got, g["stats_" + tag], rtol=1e-12)
    assert os.path.exists("cache/Toy_toy_main_dmat") and os.path.exists("cache/Toy_toy_Ds.npz")
    if not sym:
        # results file accumulates one row per run; header + format as in the reference
        ref_lines = str(g["results_csv"]).splitlines()
        lines = open("results_toy_Toy.csv").read().splitlines()
        assert lines[0] == ref_lines[0] == "name, MR, MRR, MDR, MAP,Top-1,Top-2,Top-5"
        assert lines[-1] == ref_lines[-1]
    # precomputed=True reloads what all_pairwise saved
    toy2 = Toy()
    toy2.all_pairwise(precomputed=True)
    assert np.array_equal(np.array(toy2.Ds["main"]), g["D_" + tag])
    toy.cleanup_memmap()
    assert not os.path.exists("cache/Toy_toy_main_dmat")


def test_pair_list_is_itertools_order():
    from itertools import combinations, permutations
    assert [tuple(p) for p in CoverAlgorithm.pair_list(6, True)] == list(combinations(range(6), 2))
    assert [tuple(p) for p in CoverAlgorithm.pair_list(6, False)] == list(permutations(range(6), 2))


def test_eval_statistics_equals_oracle_on_random_matrices():
    rng = np.random.default_rng(0)
    for trial in range(6):
        sizes = list(rng.integers(1, 6, size=12))
        n = int(sum(sizes))
        perm = rng.permutation(n)
        cl, p = [], 0
        for s in sizes:
            cl.append(sorted(perm[p:p + s].tolist()))
            p += s
        D = rng.random((n, n)).astype(np.float32)
        if trial % 2:
            D = np.round(D * 4) / 4          # heavy ties: the stable tie rule must agree too
        a = eval_statistics(D, cl, topsidx=(1, 3, 10), row_block=7)
        b = oracle.eval_statistics(D, cl, topsidx=(1, 3, 10), stable=True)
        np.testing.assert_allclose(np.array(a[:4]), np.array(b[:4]), rtol=1e-12)
        assert np.array_equal(a[4], b[4])


def test_bad_csv_raises_ioerror(tmp_path):
    csv = tmp_path / "bad.csv"
    csv.write_text("work_id,track_id,extra\na,b,c\n")
    with pytest.raises(IOError):
        create_dataset_filepaths(str(csv), "x/")


def test_feature_files_roundtrip(tmp_path):
    p = str(tmp_path / "w" / "t.h5")
    save_track(p, {"label": "w", "track_id": "t", "hpcp": np.ones((5, 12), np.float32),
                   "madmom_features": {"onsets": np.arange(4)}})
    d = load_track(p)
    assert d["label"] == "w" and d["hpcp"].shape == (5, 12) and list(d["madmom_features"]["onsets"]) == [0, 1, 2, 3]
    with pytest.raises(IOError):
        load_track(str(tmp_path / "w" / "missing.h5"))


def test_pool_median_equals_oracle_sync():
    rng = np.random.default_rng(1)
    for T0 in (1, 39, 40, 41, 80, 95, 1234):
        x = rng.random((T0, 12)).astype(np.float32)
        a, b = pool_median(x, 40), oracle.sync_median(x, 40)
        assert a.dtype == np.float32 and np.array_equal(a, b)


def test_serra09_class_surface(tmp_path, monkeypatch):
    """Constructor signature / attributes of the reference class (rqa_serra09.py:31-42),
    pooled feature cache, and the column-wise length normalisation -- without a GPU."""
    labels = ["a", "a", "b"]
    csv, root = _toy_dataset(tmp_path, labels)
    rng = np.random.default_rng(2)
    lens = [100, 170, 81]
    for k, l in enumerate(labels):
        save_track(root + "%s/t%d.h5" % (l, k), {"label": l, "track_id": "t%d" % k,
                                                 "hpcp": rng.random((lens[k], 12)).astype(np.float32)})
    monkeypatch.chdir(tmp_path)
    s = Serra09(csv, root, chroma_type="hpcp", shortname="toy", oti=True, kappa=0.095, tau=1, m=9, downsample_fac=40)
    assert (s.name, s.N, s.m, s.tau, s.kappa, s.downsample_fac) == ("Serra09", 3, 9, 1, 0.095, 40)
    f = s.load_features(1)
    assert f.shape == (5, 12) and s.load_features(1) is f and s.cliques == {"a": {1}}
    D = np.arange(9, dtype=np.float32).reshape(3, 3) + 1
    s.Ds["main"][:] = D
    s.normalize_by_length()
    want = oracle.serra09_normalize_by_length(D, [3, 5, 3])
    np.testing.assert_allclose(np.array(s.Ds["main"]), want, rtol=1e-7)
    p = s._params()
    assert (p.m, p.tau, p.oti) == (9, 1, 1)


def test_chenfusion_class_surface(tmp_path, monkeypatch):
    """ChenFusion (latefusion_chen.py:17-91): similarity types, the inverse length
    normalisation (a distance; unfilled cells -> inf as in the reference) and the SNF late
    fusion bookkeeping -- without a GPU."""
    from acoss_amd.algorithms.latefusion_chen import ChenFusion
    n = 24
    labels = ["w%d" % (k // 2) for k in range(n)]
    csv, root = _toy_dataset(tmp_path, labels)
    monkeypatch.chdir(tmp_path)
    c = ChenFusion(csv, root, chroma_type="hpcp", shortname="toy")
    assert c.name == "LateFusionChen" and list(c.Ds.keys()) == ["qmax", "dmax"] and c.N == n
    rng = np.random.default_rng(4)
    lens = rng.integers(40, 90, n)
    c.set_pooled_features([rng.random((int(l), 12)).astype(np.float32) for l in lens], labels)
    Q = rng.random((n, n)).astype(np.float32) * 10 + 1
    Q = np.triu(Q, 1); Q = Q + Q.T
    Dm = Q + rng.random((n, n)).astype(np.float32)
    Dm = np.triu(Dm, 1); Dm = Dm + Dm.T
    c.Ds["qmax"][:] = Q
    c.Ds["dmax"][:] = Dm
    c.normalize_by_length()
    want = np.sqrt(lens.astype(np.float64))[None, :] / np.where(Q == 0, np.nan, Q)
    got = np.array(c.Ds["qmax"])
    off = ~np.eye(n, dtype=bool)
    np.testing.assert_allclose(got[off], want[off].astype(np.float32), rtol=1e-6)
    assert np.all(np.isinf(np.diag(got)))
    # (do_late_fusion runs on the GPU only: tests/test_gpu_snf.py::test_late_fusion_of_the_host_classes)
    with pytest.raises(Exception):
        c.do_late_fusion()


def test_benchmark_rejects_unknown_algorithm(tmp_path, monkeypatch):
    import acoss_amd
    monkeypatch.chdir(tmp_path)
    with pytest.raises(NotImplementedError):
        acoss_amd.benchmark("x.csv", "feat/", algorithm="NoSuchAlgo")
    with pytest.raises(NotImplementedError):
        acoss_amd.benchmark("x.csv", "feat/", algorithm="FTM2D")
    assert "Serra09" in acoss_amd.algorithm_names and "SiMPle" in acoss_amd.algorithm_names


def test_simple_host_feature_prep_matches_reference(golden, tmp_path, monkeypatch):
    """Simple.load_features / smooth / oti (host side of a11-a12) against the goldens captured
    from the reference class (simple_silva.py:34-66)."""
    from acoss_amd.algorithms.simple_silva import Simple
    g = golden("simple")
    csv, root = _toy_dataset(tmp_path, ["w"])
    save_track(root + "w/t0.h5", {"label": "w", "track_id": "t0", "hpcp": g["feat_in"]})
    monkeypatch.chdir(tmp_path)
    s = Simple(csv, root, chroma_type="hpcp", shortname="toy")
    assert (s.name, s.SSLEN, s.WIN, s.SKIP) == ("SiMPle", 10, 200, 100)
    np.testing.assert_allclose(s.smooth(g["smooth_in"]), g["smooth_out"], rtol=1e-12, atol=1e-14)
    f = s.load_features(0)
    np.testing.assert_allclose(f, g["feat_out"], rtol=1e-10, atol=1e-12)
    assert s.load_features(0) is f                      # cached (the reference re-reads per pair)
    for k in range(6):
        Bo, order = s.oti(g["sim_A_%d" % k], g["sim_B_%d" % k])
        assert int(order[-1]) == int(g["oti_shift_%d" % k]) and np.array_equal(Bo, g["oti_B_%d" % k])


def test_snf_late_fusion_oracle_matches_reference(golden):
    """The oracle's restatement of doSimilarityFusion (the checker of acx_snf_fuse_dists) against
    the golden from the reference (similarity_fusion.py:188-196), including its aliasing of the
    work lists; the product has no host implementation (a bare doSimilarityFusion(Scores, K, niters, reg_diag) -- the
    reference's signature -- creates the process's default libacx context, which raises on a box without a GPU)."""
    from acoss_amd.algorithms.similarity_fusion import doSimilarityFusion
    g = golden("snf")
    Ws, F = oracle.snf_fuse(list(g["Ds"]), K=5, niters=4, reg_diag=1)
    np.testing.assert_allclose(np.stack(Ws), g["Ws"], rtol=1e-12)
    np.testing.assert_allclose(F, g["F"], rtol=1e-10, atol=1e-12)
    from acoss_amd import _lib
    try:
        _lib.Context(0).close()
        has_gpu = True
    except (_lib.AcxError, ImportError, OSError):
        has_gpu = False
    if not has_gpu:
        with pytest.raises((RuntimeError, ImportError, OSError)):       # AcxError is a RuntimeError: no fallback
            doSimilarityFusion(list(g["Ds"]), K=5, niters=4, reg_diag=1)


def test_earlyfusion_class_surface(tmp_path, monkeypatch):
    from acoss_amd.algorithms.earlyfusion_traile import EarlyFusion
    from acoss_amd import synth
    csv, root = _toy_dataset(tmp_path, ["a", "a", "b"])
    monkeypatch.chdir(tmp_path)
    ef = EarlyFusion(csv, root, chroma_type="hpcp", shortname="toy", blocksize=20, mfccs_per_block=50,
                     ssm_res=50, chromas_per_block=40, kappa=0.1, K=10, niters=5, log_times=False)
    assert ef.name == "EarlyFusionTraile" and sorted(ef.Ds.keys()) == ["chromas", "early", "mfccs", "ssms"]
    assert ef.get_cacheprefix() == "cache/EarlyFusionTraile_toy_hpcp"
    # the arithmetic switches of the chain: kept for the context, unknown names refused (no GPU needed for either)
    assert EarlyFusion(csv, root, shortname="toy", engine={"gemm": "bf16x3", "fuse": "exact"})._engine == {"gemm": "bf16x3", "fuse": "exact"}
    with pytest.raises(ValueError):
        EarlyFusion(csv, root, shortname="toy", engine={"arith": "f16x2"})
    with pytest.raises(KeyError):
        ef.load_features(0)                       # the toy files hold neither block nor raw MFCC / beat features
    tracks = synth.earlyfusion_set(3, seed=0, nb_range=(8, 12))
    ef.set_block_features(tracks, ["a", "a", "b"])
    assert ef.load_features(1) is tracks[1] and ef.cliques == {"a": {0, 1}, "b": {2}}
    rng = np.random.default_rng(0)
    for s in ("mfccs", "ssms", "chromas", "early"):
        D = rng.random((3, 3)) * 5
        ef.Ds[s][:] = D + D.T


def test_earlyfusion_block_feature_oracle():
    """The checker of the device's block-feature kernels (oracle.ef_resize / ef_block_features;
    earlyfusion_traile.py:100-140, 214-247).  Pinned to the reference run with the real scikit-image by
    tests/test_oracle_golden.py::test_efprep_against_reference_with_skimage; here the resize is held
    (1) to the scipy.ndimage primitives + the output clip skimage applies, and (2) to the properties of
    the algorithm: identity at scale 1, constants and linear ramps preserved, no blur when upsampling."""
    import scipy.ndimage as ndi
    rng = np.random.default_rng(11)
    X = rng.random((400, 5))
    for (i1, i2, rows) in [(30, 80, 50), (0, 240, 40), (100, 399, 50), (10, 35, 50), (0, 400, 7), (5, 6, 3)]:
        x = X[i1:i2]
        n = len(x)
        sigma = max(0.0, (n / rows - 1) / 2)
        filt = ndi.gaussian_filter(x, (sigma, 0.0), cval=0.0, mode="constant") if sigma > 0 else x
        want = ndi.zoom(filt, (rows / n, 1.0), order=1, mode="grid-constant", cval=0.0, grid_mode=True)
        want = np.where(want == 0.0, 0.0, np.clip(want, x.min(), x.max()))        # skimage warp(clip=True); 0 < min here
        np.testing.assert_allclose(oracle.ef_resize(x, rows), want, atol=1e-12)
    np.testing.assert_allclose(oracle.ef_resize(X[30:80], 50), X[30:80], atol=1e-12)          # scale 1: identity
    r = oracle.ef_resize(np.full((240, 2), 3.0), 40)
    assert r.shape == (40, 2) and np.allclose(r, 3.0)          # the zero border leaks into the first rows, the clip to [3, 3] undoes it
    r = oracle.ef_resize(np.concatenate([np.full((120, 2), -1.0), np.full((120, 2), 3.0)]), 40)
    assert r[0, 0] > -1.0 and r[-1, 0] < 3.0                    # 0 inside the range: the border does leak in
    ramp = np.arange(600, dtype=np.float64)[:, None] * np.ones((1, 2))
    r = oracle.ef_resize(ramp[100:500], 50)                                                # 8 : 1
    np.testing.assert_allclose(r[8:-8, 0], (100 + (np.arange(50) + 0.5) * 8 - 0.5)[8:-8], atol=1e-6)
    up = oracle.ef_resize(X[10:35], 50)                                                    # upsampling: no blur
    assert up.shape == (50, 5) and up.min() >= 0 and up.max() <= X[10:35].max() + 1e-12

    T = 2600
    hpcp = rng.random((T, 12)).astype(np.float32)
    mfcc = rng.standard_normal((T, 13))
    mfcc[77, 3] = np.nan
    on = np.sort(rng.choice(T - 100, 48, replace=False))
    bf = oracle.ef_block_features(hpcp, mfcc, on)
    nb = 48 - 20
    assert bf["mfccs"].shape == (nb, 650) and bf["ssms"].shape == (nb, 1225) and bf["chromas"].shape == (nb, 480)
    assert bf["mfccs"].dtype == np.float32 and np.all(np.isfinite(bf["mfccs"]))
    np.testing.assert_array_equal(bf["chroma_med"], np.median(hpcp, axis=0))
    blk = bf["mfccs"][5].reshape(50, 13).astype(np.float64)
    np.testing.assert_allclose(np.linalg.norm(blk, axis=1), 1.0, atol=1e-6)               # z-normalised rows
    np.testing.assert_allclose(blk.mean(axis=0), 0.0, atol=0.05)
    pix = np.arange(50)
    I, J = np.meshgrid(pix, pix)                                                          # the reference's D[I < J]
    np.testing.assert_allclose(bf["ssms"][5], oracle.get_csm(blk.astype(np.float32), blk.astype(np.float32))[I < J], atol=2e-3)
    np.testing.assert_allclose(bf["chromas"][7], oracle.ef_resize(hpcp[on[7]:on[27]].astype(np.float64), 40).ravel(), rtol=1e-6)




class _FakeH5(object):
    """A stand-in for the h5py module (absent here) over an in-memory tree: just enough of File /
    Group / Dataset / attrs for acoss_amd.featurestore's HDF5 branch -- deepdish layout: arrays are
    datasets, sub-dictionaries are groups, scalars and strings sit in attributes."""
    store = {}

    class Dataset(object):
        def __init__(self, a):
            self.a = np.asarray(a)

        def __getitem__(self, key):
            return self.a[key] if key != () else (self.a if self.a.ndim else self.a[()])

    class Group(object):
        def __init__(self, tree=None):
            self.tree = tree if tree is not None else {"__attrs__": {}}
            self.attrs = self.tree.setdefault("__attrs__", {})

        def items(self):
            for k, v in self.tree.items():
                if k == "__attrs__":
                    continue
                yield k, (_FakeH5.Group(v) if isinstance(v, dict) else _FakeH5.Dataset(v))

        def create_dataset(self, name, data):
            self.tree[name] = np.array(data)

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    @classmethod
    def File(cls, path, mode="r"):
        if mode == "w":
            cls.store[path] = {"__attrs__": {}}
        return cls.Group(cls.store[path])


def test_hdf5_feature_store_and_cache_branch(tmp_path, monkeypatch):
    """The .h5 paths of the feature store and of the distance-matrix cache (reference README.md:116-150,
    algorithm_template.py:90,163-166,192) through a stand-in h5py: track files in deepdish layout are
    read (nested groups, attribute scalars, byte strings), <prefix>_Ds.h5 is written next to the .npz
    and read back by all_pairwise(precomputed=True) when only the reference's file exists."""
    import sys
    from acoss_amd import featurestore
    from acoss_amd.algorithms.algorithm_template import CoverAlgorithm
    monkeypatch.setitem(sys.modules, "h5py", _FakeH5)
    rng = np.random.default_rng(0)
    labels = ["a", "a", "b"]
    csv, root = _toy_dataset(tmp_path, labels)
    for k, l in enumerate(labels):
        os.remove(root + "%s/t%d.npz" % (l, k))
        path = root + "%s/t%d.h5" % (l, k)
        open(path, "wb").close()
        _FakeH5.store[path] = {"__attrs__": {"label": l.encode(), "track_id": ("t%d" % k).encode()},
                               "hpcp": rng.random((50, 12)).astype(np.float32),
                               "madmom_features": {"__attrs__": {}, "onsets": np.arange(5)}}
    d = featurestore.load_track(root + "a/t0.h5")
    assert d["label"] == "a" and d["track_id"] == "t0" and d["hpcp"].shape == (50, 12)
    assert np.array_equal(d["madmom_features"]["onsets"], np.arange(5))
    monkeypatch.chdir(tmp_path)

    class Toy(CoverAlgorithm):
        def similarity(self, idxs):
            self.Ds["main"][idxs[:, 0], idxs[:, 1]] = 1.0 + idxs[:, 0] + 10.0 * idxs[:, 1]

    toy = Toy(csv, name="Toy", datapath=root, shortname="h5")
    toy.all_pairwise(symmetric=True)
    assert toy.cliques == {"a": {0, 1}, "b": {2}}
    want = np.array(toy.Ds["main"])
    assert os.path.exists("cache/Toy_h5_Ds.npz") and "cache/Toy_h5_Ds.h5" in _FakeH5.store
    assert np.array_equal(_FakeH5.store["cache/Toy_h5_Ds.h5"]["main"], want)
    os.remove("cache/Toy_h5_Ds.npz")
    open("cache/Toy_h5_Ds.h5", "wb").close()
    again = Toy(csv, name="Toy", datapath=root, shortname="h5")
    again.all_pairwise(symmetric=True, precomputed=True)
    assert np.array_equal(np.array(again.Ds["main"]), want)
    # a LARGE result set is cached in the reference's format only, and a stale .npz of an earlier run does not survive it
    np.savez("cache/Toy_h5_Ds.npz", main=np.zeros_like(want))
    monkeypatch.setattr(Toy, "NPZ_CACHE_BELOW", 8)
    big = Toy(csv, name="Toy", datapath=root, shortname="h5")
    big.all_pairwise(symmetric=True)
    assert not os.path.exists("cache/Toy_h5_Ds.npz")
    assert np.array_equal(_FakeH5.store["cache/Toy_h5_Ds.h5"]["main"], want)
    open("cache/Toy_h5_Ds.h5", "wb").close()
    third = Toy(csv, name="Toy", datapath=root, shortname="h5")
    third.all_pairwise(symmetric=True, precomputed=True)
    assert np.array_equal(np.array(third.Ds["main"]), want)
    # without h5py the HDF5 branch fails loudly and names the converter
    monkeypatch.setitem(sys.modules, "h5py", None)
    with pytest.raises(IOError):
        featurestore.load_track(root + "a/t0.h5")


def test_eval_statistics_counting_equals_sorting():
    """getEvalStatistics ranks a row's clique mates by counting (cliques of up to 24 songs) or by a stable argsort
    (larger cliques, rows with NaN): the same numbers, also under heavy ties -- Serra09 scores are multiples of 0.5."""
    from acoss_amd.algorithms.algorithm_template import eval_statistics
    rng = np.random.default_rng(0)
    for trial in range(12):
        n = int(rng.integers(20, 200))
        sizes = []
        while sum(sizes) < n:
            sizes.append(int(rng.integers(1, 8)))
        sizes[-1] -= sum(sizes) - n
        perm = rng.permutation(n)
        cl, p = [], 0
        for k in [k for k in sizes if k > 0]:
            cl.append([int(t) for t in perm[p:p + k]])
            p += k
        D = [rng.random((n, n)), rng.integers(0, 4, (n, n)), np.round(rng.random((n, n)) * 6) / 2][trial % 3].astype(np.float32)
        a = eval_statistics(D, cl, (1, 10, 100))
        b = eval_statistics(D, cl, (1, 10, 100), count_max_clique=0)
        assert np.allclose(a[:4], b[:4], rtol=1e-13, atol=0) and np.array_equal(a[4], b[4]), (trial, a, b)
    # against the oracle (golden-pinned to the reference's own evaluation) on a tie-free matrix
    D = rng.random((60, 60)).astype(np.float32)
    cl = [list(range(i, i + 4)) for i in range(0, 60, 4)]
    a = eval_statistics(D, cl, (1, 10))
    o = oracle.eval_statistics(D, cl, (1, 10))
    assert np.allclose(a[:4], o[:4], rtol=1e-12) and np.array_equal(a[4], o[4])


from acoss_amd.algorithms.algorithm_template import CoverAlgorithm as _CoverAlgorithm  # noqa: E402


class UserAlgo(_CoverAlgorithm):
    """Module-level (picklable by joblib's workers) user subclass in the README's style: its own CPU similarity()."""

    def similarity(self, idxs):
        for i, j in zip(idxs[:, 0], idxs[:, 1]):
            self.Ds["main"][i, j] = 100.0 * i + j + 0.5


def test_user_subclass_parallel_fanout(tmp_path, monkeypatch):
    """`all_pairwise(parallel=1, n_cores=2)` of a user subclass with its own CPU similarity(): the reference's joblib
    fan-out over the 45 chunks (algorithm_template.py:172-177) -- worker processes write through the re-opened memmaps --
    gives the serial result."""
    pytest.importorskip("joblib")
    csv, root = _toy_dataset(tmp_path, ["a", "a", "b", "b", "c", "a", "b"])
    monkeypatch.chdir(tmp_path)
    out = {}
    for par in (0, 1):
        alg = UserAlgo(csv, name="User", datapath=root, shortname="u%d" % par)
        alg.all_pairwise(parallel=par, n_cores=2, symmetric=False)
        out[par] = np.array(alg.Ds["main"])
        alg.cleanup_memmap()
    n = out[0].shape[0]
    want = np.array([[0.0 if i == j else 100.0 * i + j + 0.5 for j in range(n)] for i in range(n)], np.float32)
    assert np.array_equal(out[0], want) and np.array_equal(out[1], want)
