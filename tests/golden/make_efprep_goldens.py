#!/opt/conda/bin/python3.9
"""
Golden vectors for EarlyFusion's block features, made by THE REFERENCE ITSELF with the REAL
scikit-image: EarlyFusion.load_features and resize_block of
/root/reference/acoss/algorithms/earlyfusion_traile.py:67-154, 214-247 are imported (not restated)
and run on seeded synthetic tracks.

RUNS ONLY IN THE AUTHORING CONTAINER, under the Anaconda interpreter that ships in the image,

    /opt/conda/bin/python3.9 tests/golden/make_efprep_goldens.py

because that interpreter has scikit-image 0.18.3 (the system python3 has not).  The other packages
the reference's module pulls in are stubbed exactly as in make_goldens.py (numba.jit = identity,
deepdish = an in-memory store, librosa / progress unused here).  Output: tests/golden/efprep_skimage.npz
-- inputs and the reference's outputs, no reference source.
"""
import importlib
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(HERE, "make_goldens.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)


def main():
    import skimage
    import scipy
    mg.install_stubs()
    at = importlib.import_module("acoss.algorithms.algorithm_template")
    ef = importlib.import_module("acoss.algorithms.earlyfusion_traile")
    out = {"versions": np.array(["skimage %s" % skimage.__version__, "numpy %s" % np.__version__, "scipy %s" % scipy.__version__])}
    rng = np.random.default_rng(20261002)

    # ---- resize_block alone (earlyfusion_traile.py:214-247): down- and up-sampling, the identity, tiny blocks
    X = rng.standard_normal((900, 13)).astype(np.float32)
    C = rng.random((900, 12)).astype(np.float32)
    cases = [("mfcc", 0, 400, 50), ("mfcc", 100, 151, 50), ("mfcc", 300, 350, 50), ("mfcc", 10, 37, 50), ("mfcc", 5, 8, 50),
             ("mfcc", 0, 900, 50), ("mfcc", 200, 323, 32), ("chroma", 50, 470, 40), ("chroma", 600, 640, 40), ("chroma", 700, 713, 40),
             ("chroma", 0, 899, 24), ("mfcc", 420, 421, 50)]
    out["rb_X"], out["rb_C"] = X, C
    out["rb_cases"] = np.array([[0 if c[0] == "mfcc" else 1, c[1], c[2], c[3]] for c in cases], np.int64)
    for k, (which, i1, i2, rows) in enumerate(cases):
        src = X if which == "mfcc" else C
        out["rb_out_%d" % k] = ef.resize_block(src, i1, i2, rows)

    # ---- the whole of EarlyFusion.load_features on synthetic tracks
    def track(T, nbeats, nan=False, ncoef=13):
        hpcp = rng.random((T, 12)).astype(np.float32)
        mfcc = rng.standard_normal((ncoef, T)).astype(np.float32)            # feats['mfcc_htk'] is (ncoef, T)
        if nan:
            mfcc[3, T // 3] = np.nan
        on = np.sort(rng.choice(T - 10, nbeats, replace=False)).astype(np.int64)
        return {"hpcp": hpcp, "mfcc_htk": mfcc, "madmom_features": {"onsets": on}, "label": "w"}

    tracks = [track(2600, 40, nan=True),         # blocks of ~ 1200 frames: 24 : 1 down-sampling
              track(420, 34),                    # ~ 12 frames per beat: blocks of ~ 230 frames
              track(140, 36)]                    # blocks shorter than 50 rows: up-sampling
    store = {}
    at.CoverAlgorithm.load_features = lambda self, i: {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v)
                                                       for k, v in store[i].items()}
    algo = object.__new__(ef.EarlyFusion)
    algo.chroma_type, algo.blocksize, algo.mfccs_per_block, algo.chromas_per_block = "hpcp", 20, 50, 40
    algo.all_block_feats, algo.log_times = {}, False
    algo.get_cacheprefix = lambda: "/nonexistent/efprep"
    ef.dd.io.save = lambda path, obj: None
    for k, t in enumerate(tracks):
        store[k] = t
        bf = algo.load_features(k)
        out["lf%d_hpcp" % k] = t["hpcp"]
        out["lf%d_mfcc_htk" % k] = t["mfcc_htk"]
        out["lf%d_onsets" % k] = t["madmom_features"]["onsets"]
        for key in ("mfccs", "ssms", "chromas", "chroma_med"):
            out["lf%d_%s" % (k, key)] = np.asarray(bf[key])
    # other block geometry (ctor arguments blocksize / mfccs_per_block / chromas_per_block)
    algo2 = object.__new__(ef.EarlyFusion)
    algo2.chroma_type, algo2.blocksize, algo2.mfccs_per_block, algo2.chromas_per_block = "hpcp", 12, 32, 24
    algo2.all_block_feats, algo2.log_times = {}, False
    algo2.get_cacheprefix = lambda: "/nonexistent/efprep2"
    store[9] = track(1000, 30, ncoef=20)
    bf = algo2.load_features(9)
    out["lf9_hpcp"], out["lf9_mfcc_htk"], out["lf9_onsets"] = store[9]["hpcp"], store[9]["mfcc_htk"], store[9]["madmom_features"]["onsets"]
    for key in ("mfccs", "ssms", "chromas", "chroma_med"):
        out["lf9_%s" % key] = np.asarray(bf[key])
    path = os.path.join(HERE, "efprep_skimage.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", ", ".join(out["versions"]))


if __name__ == "__main__":
    main()
