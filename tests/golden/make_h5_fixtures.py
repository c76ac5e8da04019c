#!/opt/conda/bin/python3.9
"""
HDF5 fixtures written by the REAL PyTables (3.6.1, on libhdf5 1.10.6) in the layout deepdish gives its
files -- the reference's per-track feature files (README.md:116-150; dd.io.save in extractors.py) and its
<prefix>_Ds.h5 matrix cache (algorithm_template.py:192).

RUNS ONLY IN THE AUTHORING CONTAINER under the image's Anaconda interpreter:

    /opt/conda/bin/python3.9 tests/golden/make_h5_fixtures.py

deepdish itself is not installed anywhere in the image; save_like_deepdish() below follows what
deepdish 0.3.6's hdf5io.save does level by level, on the real PyTables API:
    dict           -> group, TITLE "dict:<n>"        list / tuple -> group "list:<n>" / "tuple:<n>" of i0, i1, ...
    ndarray        -> CArray (zlib 9 + shuffle) when it has more than 300 elements, else Array;
                      empty arrays as their shape + attribute zeroarray_dtype; unicode arrays as uint8 + strtype / itemsize
    str, int, float, bool, numpy scalars -> attributes of the enclosing group;    None -> empty group "nonetype:"
(PyTables 3.6.1 predates numpy 1.24: the removed aliases it imports are put back first.)
Outputs (tests/golden/h5/): track_deepdish.h5, Ds_deepdish.h5 and expected.npz with the same content.
"""
import os
import sys

import numpy

numpy.typeDict = numpy.sctypeDict
for _n in ("float", "int", "bool", "object", "complex", "str"):
    if not hasattr(numpy, _n):
        setattr(numpy, _n, getattr(__builtins__, _n))
import numpy as np  # noqa: E402
import tables  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "h5")
ATTR_TYPES = (int, float, bool, str, bytes, np.generic)


def _save_level(h, group, level, name, filters):
    if isinstance(level, dict):
        g = h.create_group(group, name, "dict:%d" % len(level))
        for k, v in level.items():
            _save_level(h, g, v, k, filters)
    elif isinstance(level, (list, tuple)):
        g = h.create_group(group, name, "%s:%d" % ("list" if isinstance(level, list) else "tuple", len(level)))
        for i, v in enumerate(level):
            _save_level(h, g, v, "i%d" % i, filters)
    elif isinstance(level, np.ndarray):
        x = level
        strtype = itemsize = None
        if x.dtype.kind == "U":
            strtype, itemsize = b"unicode", x.itemsize // 4
            x = x.view(dtype=np.uint8)
        if x.ndim > 0 and np.min(x.shape) == 0:
            sh = np.array(x.shape, np.int64)
            node = h.create_array(group, name, atom=tables.Int64Atom(), shape=(sh.size,))
            node._v_attrs.zeroarray_dtype = np.dtype(level.dtype).str.encode("ascii")
            node[:] = sh
            return
        if x.ndim == 0:
            setattr(group._v_attrs, name, x[()])
            return
        atom = tables.Atom.from_dtype(x.dtype)
        if filters is not None and x.size > 300:
            node = h.create_carray(group, name, atom=atom, shape=x.shape, filters=filters)
        else:
            node = h.create_array(group, name, atom=atom, shape=x.shape)
        if strtype is not None:
            node._v_attrs.strtype = strtype
            node._v_attrs.itemsize = itemsize
        node[:] = x
    elif isinstance(level, ATTR_TYPES):
        setattr(group._v_attrs, name, level)
    elif level is None:
        h.create_group(group, name, "nonetype:")
    else:
        raise TypeError(type(level))


def save_like_deepdish(path, data):
    filters = tables.Filters(complevel=9, complib="zlib", shuffle=True)          # deepdish's default ('zlib', level 9)
    with tables.open_file(path, mode="w") as h:
        h.root._v_attrs["DEEPDISH_IO_VERSION"] = 12
        for k, v in data.items():
            _save_level(h, h.root, v, k, filters)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)
    T = 57
    track = {
        "hpcp": rng.random((T, 12)).astype(np.float32), "crema": rng.random((T, 12)).astype(np.float32),
        "chroma_cens": rng.random((T, 12)),                                            # f64 like librosa's
        "mfcc_htk": rng.standard_normal((13, T)).astype(np.float32),
        "madmom_features": {"onsets": np.sort(rng.choice(T, 9, replace=False)).astype(np.int64), "tempos": rng.random((2, 2)),
                            "novfn": rng.random(T).astype(np.float32), "snovfn": rng.random(T).astype(np.float32)},
        "key_extractor": {"key": "C#", "scale": "minor", "strength": 0.7312},
        "tags": [("artist", "somebody"), ("title", "some song é")],
        "label": "W_163", "track_id": "P_163_1", "duration": 201.5, "n_frames": T, "is_cover": True,
        "empty": np.zeros((0, 12), np.float32), "names": np.array(["ab", "cdeü"]), "nothing": None,
    }
    save_like_deepdish(os.path.join(OUT, "track_deepdish.h5"), track)
    Ds = {"main": rng.random((23, 23)).astype(np.float32), "qmax": rng.random((7, 7)).astype(np.float32),
          "late": rng.random((23, 23))}
    save_like_deepdish(os.path.join(OUT, "Ds_deepdish.h5"), Ds)
    flat = {}
    for k, v in track.items():
        if isinstance(v, dict):
            for k2, v2 in v.items():
                flat["track/%s/%s" % (k, k2)] = np.asarray(v2)
        elif k == "tags":
            flat["track/tags"] = np.array(v)
        elif v is None:
            continue
        else:
            flat["track/%s" % k] = np.asarray(v)
    for k, v in Ds.items():
        flat["Ds/%s" % k] = v
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **flat)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
    print("tables", tables.__version__, "hdf5", tables.hdf5_version, "numpy", np.__version__)


if __name__ == "__main__":
    main()
