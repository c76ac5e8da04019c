"""
Per-track feature files and the distance-matrix cache.  The reference stores one deepdish HDF5 per track
at feature_dir/work_id/track_id.h5 (README.md:116-150, algorithm_template.py:90) and the matrices in
<prefix>_Ds.h5 (:163-166, :192).  Both are read and written here through h5py when it is importable, else
through the HDF5 C library itself (acoss_amd.hdf5: libhdf5 via ctypes -- the image has the library but no
Python binding for the system interpreter); without either, HDF5 files raise IOError.  A `.npz` next to a
track's `.h5` with the same stem takes precedence (same keys: hpcp / crema (T,12), mfcc_htk, label,
track_id, madmom_features_onsets ...): the format of the synthetic sets and of scripts/h5_to_npz.py.
"""
import os

import numpy as np

__all__ = ["load_track", "save_track", "save_matrices_h5", "load_matrices_h5"]


def hdf5_backend():
    """'h5py', 'libhdf5' (ctypes) or None."""
    try:
        import h5py  # noqa: F401
        return "h5py"
    except ImportError:
        pass
    from . import hdf5
    return "libhdf5" if hdf5.available() else None


def _load_h5(path):
    backend = hdf5_backend()
    if backend is None:
        raise IOError("%s is an HDF5 file but neither h5py nor the HDF5 C library is available; convert the "
                      "features to .npz (acoss_amd.featurestore.save_track)" % path)
    if backend == "libhdf5":
        from . import hdf5
        d = hdf5.read_tree(path)
        for k in ("label", "track_id"):
            if isinstance(d.get(k), bytes):
                d[k] = d[k].decode()
        return d
    import h5py

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


def save_track(path, feats, fmt="npz"):
    """Write a feature dict next to `path` (any extension): fmt="npz" (default) or "h5" -- the reference's own
    per-track layout (deepdish: arrays as datasets, sub-dictionaries as groups, strings / scalars as
    attributes), through the HDF5 C library."""
    stem, _ = os.path.splitext(path)
    os.makedirs(os.path.dirname(stem) or ".", exist_ok=True)
    if fmt == "h5":
        from . import hdf5
        hdf5.write_tree(stem + ".h5", feats)
        return
    if fmt != "npz":
        raise ValueError("save_track: fmt must be 'npz' or 'h5'")
    flat = {}
    for k, v in feats.items():
        if isinstance(v, dict):
            for k2, v2 in v.items():
                flat["%s_%s" % (k, k2)] = np.asarray(v2)
        else:
            flat[k] = np.asarray(v)
    np.savez(stem + ".npz", **flat)


# matrices above this size go to the cache uncompressed: zlib makes ~40 MB/s of float scores and saves a fifth of the file --
# 24 s per 15 000 x 15 000 plane, six planes for EarlyFusion, inside all_pairwise (measured: scripts/end_to_end*.py)
H5_COMPRESS_BELOW = 64 << 20


def save_matrices_h5(path, Ds):
    """{name: (N, N) array} -> one HDF5 file with a dataset per name -- what the reference's
    dd.io.save("<prefix>_Ds.h5", self.Ds) leaves on disk (algorithm_template.py:192; deepdish stores
    arrays of more than 300 elements as chunked, shuffled, compressed CArrays; here zlib up to 64 MB per
    matrix, plain chunks above -- either reads back through dd.io.load / load_matrices_h5).  Returns False when no
    HDF5 backend exists (the caller then writes the .npz cache)."""
    backend = hdf5_backend()
    if backend is None:
        return False
    if backend == "libhdf5":
        from . import hdf5
        big = any(np.asarray(v).nbytes > H5_COMPRESS_BELOW for v in Ds.values())
        hdf5.write_tree(path, {k: np.asarray(v) for k, v in Ds.items()}, compress=0 if big else 1)
        return True
    import h5py
    with h5py.File(path, "w") as f:
        for k, v in Ds.items():
            f.create_dataset(k, data=np.asarray(v))
    return True


def load_matrices_h5(path):
    """The reference's <prefix>_Ds.h5 (dd.io.load, algorithm_template.py:163-166) -> {name: array}."""
    d = _load_h5(path)
    return {k: np.asarray(v) for k, v in d.items() if isinstance(v, np.ndarray) or np.ndim(v) == 2}
