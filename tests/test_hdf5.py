"""
CPU suite: the reference's on-disk formats through a REAL HDF5 library.

The reference keeps one deepdish file per track (README.md:116-150; algorithm_template.py:90) and the
distance matrices in <prefix>_Ds.h5 (:163-166, :192).  deepdish = PyTables = libhdf5.  This image has
the C library (/opt/conda/lib/libhdf5.so 1.10.6) but no Python binding for the system interpreter, so
acoss_amd.hdf5 binds it with ctypes.  Evidence, in both directions:
  * tests/golden/h5/*.h5 were WRITTEN BY THE REAL PyTables 3.6.1 in deepdish's layout
    (tests/golden/make_h5_fixtures.py, run under the image's Anaconda python3.9) and are read here by
    the product code (featurestore.load_track / load_matrices_h5);
  * files WRITTEN by the product code (save_matrices_h5, save_track(fmt="h5")) are read back by the
    product, by the HDF5 distribution's own h5dump, and by the real h5py (subprocess under
    /opt/conda/bin/python3.9) when those exist;
  * the host classes run over `.h5` track files and a `<prefix>_Ds.h5` cache end to end.
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5 = os.path.join(ROOT, "tests", "golden", "h5")
CONDA_PY = "/opt/conda/bin/python3.9"


def _need_lib():
    from acoss_amd import hdf5
    if not hdf5.available():
        pytest.skip("no HDF5 C library (libhdf5) on this machine")
    return hdf5


def test_reads_files_written_by_pytables():
    _need_lib()
    from acoss_amd import featurestore as fs
    e = np.load(os.path.join(H5, "expected.npz"))
    t = fs.load_track(os.path.join(H5, "track_deepdish.h5"))
    for key in ("hpcp", "crema", "chroma_cens", "mfcc_htk", "empty"):
        assert t[key].dtype == e["track/" + key].dtype and np.array_equal(t[key], e["track/" + key]), key
    for key in ("onsets", "tempos", "novfn", "snovfn"):
        assert np.array_equal(t["madmom_features"][key], e["track/madmom_features/" + key]), key
    assert t["madmom_features"]["onsets"].dtype == np.int64
    assert t["label"] == "W_163" and t["track_id"] == "P_163_1" and isinstance(t["label"], str)
    assert t["key_extractor"] == {"key": "C#", "scale": "minor", "strength": 0.7312}
    assert t["duration"] == 201.5 and t["n_frames"] == 57 and t["is_cover"] is True or t["is_cover"] == True  # noqa: E712
    assert t["tags"] == [("artist", "somebody"), ("title", "some song é")]
    assert list(t["names"]) == ["ab", "cdeü"] and t["nothing"] is None
    assert "CLASS" not in t and "TITLE" not in t and "DEEPDISH_IO_VERSION" not in t
    D = fs.load_matrices_h5(os.path.join(H5, "Ds_deepdish.h5"))
    assert sorted(D) == ["late", "main", "qmax"]
    for k in D:
        assert D[k].dtype == e["Ds/" + k].dtype and np.array_equal(D[k], e["Ds/" + k])


def test_written_files_are_real_hdf5(tmp_path, monkeypatch):
    hdf5 = _need_lib()
    from acoss_amd import featurestore as fs
    rng = np.random.default_rng(1)
    Ds = {"main": rng.random((40, 40)).astype(np.float32), "qmax": rng.random((5, 5)).astype(np.float32), "Late": rng.random((40, 40))}
    p = str(tmp_path / "X_Ds.h5")
    assert fs.save_matrices_h5(p, Ds) is True
    back = fs.load_matrices_h5(p)
    assert sorted(back) == sorted(Ds)
    for k in Ds:
        assert back[k].dtype == Ds[k].dtype and np.array_equal(back[k], Ds[k])
    # matrices above the compression limit are stored in plain chunks (zlib makes ~40 MB/s of float scores) and read back the same
    monkeypatch.setattr(fs, "H5_COMPRESS_BELOW", 1000)
    p2 = str(tmp_path / "Y_Ds.h5")
    assert fs.save_matrices_h5(p2, Ds) is True
    back2 = fs.load_matrices_h5(p2)
    for k in Ds:
        assert back2[k].dtype == Ds[k].dtype and np.array_equal(back2[k], Ds[k])
    assert os.path.getsize(p2) >= sum(v.nbytes for v in Ds.values())
    feats = {"hpcp": rng.random((77, 12)).astype(np.float32), "mfcc_htk": rng.standard_normal((13, 77)).astype(np.float32),
             "madmom_features": {"onsets": np.arange(0, 70, 7, dtype=np.int64)}, "label": "W_9", "track_id": "P_9_2", "duration": 12.5}
    fs.save_track(str(tmp_path / "w/t.h5"), feats, fmt="h5")
    t = fs.load_track(str(tmp_path / "w/t.h5"))
    assert t["label"] == "W_9" and t["duration"] == 12.5 and np.array_equal(t["hpcp"], feats["hpcp"])
    assert np.array_equal(t["madmom_features"]["onsets"], feats["madmom_features"]["onsets"])
    # the HDF5 distribution's own tool parses the file: 1600 elements > 300 -> chunked + deflate, 25 -> contiguous
    dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(dump):
        r = subprocess.run([dump, "-H", "-p", p], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert 'DATASET "main"' in r.stdout and "H5T_IEEE_F32LE" in r.stdout and "H5T_IEEE_F64LE" in r.stdout
        assert "COMPRESSION DEFLATE" in r.stdout and "CHUNKED" in r.stdout and "CONTIGUOUS" in r.stdout
    # the real h5py reads what we wrote
    if os.path.exists(CONDA_PY):
        code = ("import h5py, json, numpy as np, sys\n"
                "f = h5py.File(sys.argv[1], 'r')\n"
                "print(json.dumps({k: [list(f[k].shape), str(f[k].dtype), float(np.asarray(f[k]).sum()), f[k].compression] for k in f}))\n"
                "g = h5py.File(sys.argv[2], 'r')\n"
                "print(json.dumps({'label': g.attrs['label'].decode() if isinstance(g.attrs['label'], bytes) else str(g.attrs['label']),\n"
                "                  'onsets': [int(v) for v in g['madmom_features/onsets'][()]], 'title': g['madmom_features'].attrs['TITLE'].decode()}))\n")
        r = subprocess.run([CONDA_PY, "-W", "ignore", "-c", code, p, str(tmp_path / "w/t.h5")], capture_output=True, text=True)
        if r.returncode != 0 and "No module named" in r.stderr:
            pytest.skip("h5py is not installed for %s" % CONDA_PY)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.strip().splitlines()
        got = json.loads(lines[0])
        for k in Ds:
            assert got[k][0] == list(Ds[k].shape) and got[k][1] == str(Ds[k].dtype)
            assert abs(got[k][2] - float(np.asarray(Ds[k], np.float64).sum())) <= 1e-3
        assert got["main"][3] == "gzip" and got["qmax"][3] is None
        meta = json.loads(lines[1])
        assert meta == {"label": "W_9", "onsets": list(range(0, 70, 7)), "title": "dict:1"}
    assert hdf5.library_version().split(".")[0] == "1"


def test_host_classes_over_hdf5_files(tmp_path, monkeypatch):
    """A user subclass of CoverAlgorithm over `.h5` track files (no .npz anywhere): labels come from the files'
    attributes, all_pairwise leaves <prefix>_Ds.h5 next to the .npz cache, and precomputed=True restores the
    matrices from the HDF5 file alone -- the reference's own round trip (algorithm_template.py:163-166, 192)."""
    _need_lib()
    from acoss_amd import featurestore as fs
    from acoss_amd.algorithms.algorithm_template import CoverAlgorithm
    from acoss_amd.algorithms.rqa_serra09 import Serra09
    rng = np.random.default_rng(3)
    labels = ["a", "a", "b", "b", "b", "c"]
    root = str(tmp_path / "feat") + "/"
    with open(tmp_path / "ds.csv", "w") as f:
        f.write("work_id,track_id\n")
        for k, l in enumerate(labels):
            fs.save_track(root + "%s/t%d.h5" % (l, k), {"label": l, "track_id": "t%d" % k,
                                                        "hpcp": rng.random((200 + 10 * k, 12)).astype(np.float32)}, fmt="h5")
            f.write("%s,t%d\n" % (l, k))
    assert not any(fn.endswith(".npz") for _, _, fns in os.walk(root) for fn in fns)
    monkeypatch.chdir(tmp_path)
    S = rng.random((6, 6)).astype(np.float32)

    class Toy(CoverAlgorithm):
        def similarity(self, idxs):
            self.Ds["main"][idxs[:, 0], idxs[:, 1]] = S[idxs[:, 0], idxs[:, 1]]

    toy = Toy(str(tmp_path / "ds.csv"), name="Toy", datapath=root, shortname="h5")
    toy.all_pairwise(symmetric=True)
    assert toy.cliques == {"a": {0, 1}, "b": {2, 3, 4}, "c": {5}}
    want = np.array(toy.Ds["main"])
    prefix = toy.get_cacheprefix()
    assert os.path.exists(prefix + "_Ds.h5") and os.path.exists(prefix + "_Ds.npz")
    os.remove(prefix + "_Ds.npz")
    again = Toy(str(tmp_path / "ds.csv"), name="Toy", datapath=root, shortname="h5")
    again.all_pairwise(symmetric=True, precomputed=True)
    assert np.array_equal(np.array(again.Ds["main"]), want)
    # Serra09.load_features pools the chroma it finds in the HDF5 file (x40 median, rqa_serra09.py:44-53)
    s9 = Serra09(str(tmp_path / "ds.csv"), root, shortname="h5s")
    x = s9.load_features(4)
    raw = fs.load_track(root + "b/t4.h5")["hpcp"]
    assert x.shape == (6, 12) and np.array_equal(x[0], np.median(raw[:40], axis=0))
    s9.cleanup_memmap(); toy.cleanup_memmap(); again.cleanup_memmap()
