"""
Per-track feature files.  The reference stores one deepdish HDF5 per track at
feature_dir/work_id/track_id.h5 (README.md:116-150, algorithm_template.py:90).  Neither
h5py nor deepdish exists in the build environment, so the native format here is one
`.npz` per track at the same place (same keys: hpcp / crema (T,12), mfcc_htk, label,
track_id, madmom_features_onsets ...).  `.h5` files are read when h5py is importable.
"""
import os

import numpy as np

__all__ = ["load_track", "save_track", "save_matrices_h5", "load_matrices_h5"]


def _load_h5(path):
    try:
        import h5py
    except ImportError:
        raise IOError("%s is an HDF5 file but h5py is not installed; convert the features to .npz "
                      "(acoss_amd.featurestore.save_track)" % path)

    def rec(g):
        out = {}
        for k, v in g.items():
            out[k] = rec(v) if isinstance(v, h5py.Group) else v[()]
        for k, v in g.attrs.items():       # deepdish keeps scalars / strings as attributes
            out.setdefault(k, v)
        return out
    with h5py.File(path, "r") as f:
        d = rec(f)
    for k in ("label", "track_id"):
        if isinstance(d.get(k), bytes):
            d[k] = d[k].decode()
    return d


def load_track(path):
    """Feature dict of one track.  `path` is what create_dataset_filepaths built (".h5");
    an ".npz" next to it with the same stem takes precedence."""
    stem, ext = os.path.splitext(path)
    npz = stem + ".npz"
    if os.path.exists(npz):
        with np.load(npz, allow_pickle=False) as z:
            d = {k: z[k] for k in z.files}
        for k in ("label", "track_id"):
            if k in d:
                d[k] = str(d[k])
        if "madmom_features_onsets" in d:
            d["madmom_features"] = {"onsets": d.pop("madmom_features_onsets")}
        return d
    if os.path.exists(path) and ext in (".h5", ".hdf5"):
        return _load_h5(path)
    raise IOError("feature file not found: %s (or %s)" % (path, npz))


def save_track(path, feats):
    """Write a feature dict as .npz next to `path` (any extension)."""
    stem, _ = os.path.splitext(path)
    os.makedirs(os.path.dirname(stem) or ".", exist_ok=True)
    flat = {}
    for k, v in feats.items():
        if isinstance(v, dict):
            for k2, v2 in v.items():
                flat["%s_%s" % (k, k2)] = np.asarray(v2)
        else:
            flat[k] = np.asarray(v)
    np.savez(stem + ".npz", **flat)


def save_matrices_h5(path, Ds):
    """{name: (N, N) array} -> one HDF5 file with a dataset per name -- what the reference's
    dd.io.save("<prefix>_Ds.h5", self.Ds) leaves on disk (algorithm_template.py:192; deepdish stores
    plain ndarrays as plain datasets).  Returns False when h5py is not installed (the .npz cache is
    always written)."""
    try:
        import h5py
    except ImportError:
        return False
    with h5py.File(path, "w") as f:
        for k, v in Ds.items():
            f.create_dataset(k, data=np.asarray(v))
    return True


def load_matrices_h5(path):
    """The reference's <prefix>_Ds.h5 (dd.io.load, algorithm_template.py:163-166) -> {name: array}."""
    d = _load_h5(path)
    return {k: np.asarray(v) for k, v in d.items() if isinstance(v, np.ndarray) or np.ndim(v) == 2}
