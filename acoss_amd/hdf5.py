"""
HDF5 files through the HDF5 C library itself (libhdf5 via ctypes) -- for environments where the
reference's I/O stack (deepdish -> PyTables -> libhdf5) is not installed but the C library is (this
image: /opt/conda/lib/libhdf5.so, no h5py / tables / deepdish for the system interpreter).

What the reference stores (README.md:116-150, algorithm_template.py:90,163-166,192): one deepdish file
per track -- a dictionary saved by PyTables: numeric arrays as (chunked, zlib-compressed) datasets,
sub-dictionaries as groups, strings and scalars as attributes of their group -- and the distance
matrices as one dataset per similarity type in <prefix>_Ds.h5.

    read_tree(path)                      -> nested dict: groups -> dicts, datasets -> numpy arrays (numeric,
                                            bool, fixed / variable length strings), attributes -> entries of
                                            their group's dict (PyTables' own bookkeeping attributes dropped)
    write_tree(path, tree, compress=..)  the inverse, in the layout PyTables / deepdish produce: arrays of more
                                            than 300 elements chunked + shuffle + deflate, str / int / float /
                                            bool scalars as attributes, dicts as groups

Only what those files need: no compound types, no references, no partial I/O.
"""
import ctypes
import ctypes.util
import os

import numpy as np

__all__ = ["available", "read_tree", "write_tree", "library_version"]

_hid = ctypes.c_int64
_hsize = ctypes.c_uint64
_L = None
_CANDIDATES = ("libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so",
               "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so")

# PyTables / deepdish bookkeeping that is not data
_SYSTEM_ATTRS = {"CLASS", "VERSION", "TITLE", "FLAVOR", "PYTABLES_FORMAT_VERSION", "FILTERS", "DEEPDISH_IO_VERSION",
                 "DEEPDISH_IO_UNPACK"}

H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_BITFIELD, H5T_ENUM = 0, 1, 3, 4, 8
H5I_GROUP, H5I_DATASET = 2, 5


class _GInfo(ctypes.Structure):
    _fields_ = [("storage_type", ctypes.c_int), ("nlinks", _hsize), ("max_corder", ctypes.c_int64), ("mounted", ctypes.c_int)]


def _lib():
    global _L
    if _L is not None:
        return _L
    names = [os.environ["ACX_HDF5_LIB"]] if os.environ.get("ACX_HDF5_LIB") else []
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        names.append(found)
    names += list(_CANDIDATES)
    last = None
    for n in names:
        try:
            L = ctypes.CDLL(n)
            break
        except OSError as e:
            last = e
    else:
        raise ImportError("the HDF5 C library (libhdf5) was not found (set ACX_HDF5_LIB to its path): %s" % last)
    if L.H5open() < 0:
        raise ImportError("H5open failed")
    L.H5Eset_auto2.argtypes = [_hid, ctypes.c_void_p, ctypes.c_void_p]
    L.H5Eset_auto2(0, None, None)                     # no error stack on stderr: failures come back as return codes
    sig = {
        "H5Fopen": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]), "H5Fcreate": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
        "H5Fclose": (ctypes.c_int, [_hid]), "H5Gget_info": (ctypes.c_int, [_hid, ctypes.POINTER(_GInfo)]),
        "H5Gcreate2": (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid]), "H5Gclose": (ctypes.c_int, [_hid]),
        "H5Lget_name_by_idx": (ctypes.c_ssize_t, [_hid, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, _hsize, ctypes.c_char_p,
                                                  ctypes.c_size_t, _hid]),
        "H5Oopen": (_hid, [_hid, ctypes.c_char_p, _hid]), "H5Oclose": (ctypes.c_int, [_hid]), "H5Iget_type": (ctypes.c_int, [_hid]),
        "H5Dget_type": (_hid, [_hid]), "H5Dget_space": (_hid, [_hid]),
        "H5Dread": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
        "H5Dwrite": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
        "H5Dcreate2": (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]), "H5Dclose": (ctypes.c_int, [_hid]),
        "H5Dvlen_reclaim": (ctypes.c_int, [_hid, _hid, _hid, ctypes.c_void_p]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [_hid, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
        "H5Sget_simple_extent_type": (ctypes.c_int, [_hid]),
        "H5Screate_simple": (_hid, [ctypes.c_int, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]), "H5Screate": (_hid, [ctypes.c_int]),
        "H5Sclose": (ctypes.c_int, [_hid]),
        "H5Tget_class": (ctypes.c_int, [_hid]), "H5Tget_size": (ctypes.c_size_t, [_hid]), "H5Tget_sign": (ctypes.c_int, [_hid]),
        "H5Tis_variable_str": (ctypes.c_int, [_hid]), "H5Tget_super": (_hid, [_hid]), "H5Tcopy": (_hid, [_hid]),
        "H5Tset_size": (ctypes.c_int, [_hid, ctypes.c_size_t]), "H5Tset_cset": (ctypes.c_int, [_hid, ctypes.c_int]),
        "H5Tget_cset": (ctypes.c_int, [_hid]), "H5Tclose": (ctypes.c_int, [_hid]),
        "H5Aopen_by_idx": (_hid, [_hid, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, _hsize, _hid, _hid]),
        "H5Aget_name": (ctypes.c_ssize_t, [_hid, ctypes.c_size_t, ctypes.c_char_p]), "H5Aget_type": (_hid, [_hid]),
        "H5Aget_space": (_hid, [_hid]), "H5Aread": (ctypes.c_int, [_hid, _hid, ctypes.c_void_p]),
        "H5Acreate2": (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid]),
        "H5Awrite": (ctypes.c_int, [_hid, _hid, ctypes.c_void_p]), "H5Aclose": (ctypes.c_int, [_hid]),
        "H5Pcreate": (_hid, [_hid]), "H5Pset_chunk": (ctypes.c_int, [_hid, ctypes.c_int, ctypes.POINTER(_hsize)]),
        "H5Pset_shuffle": (ctypes.c_int, [_hid]), "H5Pset_deflate": (ctypes.c_int, [_hid, ctypes.c_uint]),
        "H5Pclose": (ctypes.c_int, [_hid]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _L = L
    return L


def available():
    try:
        _lib()
        return True
    except (ImportError, OSError, AttributeError):
        return False


def library_version():
    L = _lib()
    a, b, c = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    L.H5get_libversion(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return "%d.%d.%d" % (a.value, b.value, c.value)


def _g(name):
    return _hid.in_dll(_lib(), name).value


def _native(dtype):
    dtype = np.dtype(dtype)
    table = {"f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g", "i1": "H5T_NATIVE_INT8_g", "u1": "H5T_NATIVE_UINT8_g",
             "i2": "H5T_NATIVE_INT16_g", "u2": "H5T_NATIVE_UINT16_g", "i4": "H5T_NATIVE_INT32_g", "u4": "H5T_NATIVE_UINT32_g",
             "i8": "H5T_NATIVE_INT64_g", "u8": "H5T_NATIVE_UINT64_g"}
    key = dtype.kind + str(dtype.itemsize)
    if key not in table:
        raise TypeError("HDF5: unsupported dtype %s" % dtype)
    return _g(table[key])


def _shape(space):
    L = _lib()
    nd = L.H5Sget_simple_extent_ndims(space)
    if nd <= 0:
        return ()
    dims = (_hsize * nd)()
    L.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(d) for d in dims)


def _read(obj, ftype, space, reader):
    """Value of a dataset / attribute: reader(memtype, buffer) performs the H5Dread / H5Aread."""
    L = _lib()
    shape = _shape(space)
    n = int(np.prod(shape)) if shape else 1
    if L.H5Sget_simple_extent_type(space) == 2:        # H5S_NULL: no data (PyTables' empty TITLE)
        return None
    cls = L.H5Tget_class(ftype)
    size = L.H5Tget_size(ftype)
    if cls in (H5T_INTEGER, H5T_FLOAT, H5T_BITFIELD, H5T_ENUM):
        base = ftype
        sup = None
        if cls == H5T_ENUM:                           # PyTables / h5py booleans: an 8-bit enum
            sup = L.H5Tget_super(ftype)
            base = sup
            size = L.H5Tget_size(base)
        if cls == H5T_FLOAT:
            dt = np.dtype("f%d" % size)
        elif cls == H5T_BITFIELD:
            dt = np.dtype("u%d" % size)
        else:
            dt = np.dtype(("i%d" if L.H5Tget_sign(base) == 1 else "u%d") % size)
        out = np.empty(shape, dt)
        # (bit fields -- PyTables' booleans are H5T_STD_B8 -- convert to nothing but bit fields)
        mem = _g("H5T_NATIVE_B%d_g" % (8 * size)) if cls == H5T_BITFIELD else _native(dt)
        rc = reader(mem, out.ctypes.data_as(ctypes.c_void_p))
        if sup is not None:
            L.H5Tclose(sup)
        if rc < 0:
            raise IOError("HDF5: read failed")
        if cls in (H5T_ENUM, H5T_BITFIELD) and size == 1:
            out = out.astype(bool)
        return out if shape else out[()]
    if cls == H5T_STRING:
        utf8 = L.H5Tget_cset(ftype) == 1
        if L.H5Tis_variable_str(ftype) > 0:
            mt = L.H5Tcopy(_g("H5T_C_S1_g"))
            L.H5Tset_size(mt, ctypes.c_size_t(-1).value)
            L.H5Tset_cset(mt, L.H5Tget_cset(ftype))
            buf = (ctypes.c_char_p * n)()
            if reader(mt, ctypes.cast(buf, ctypes.c_void_p)) < 0:
                L.H5Tclose(mt)
                raise IOError("HDF5: read failed")
            vals = [(b or b"").decode("utf-8" if utf8 else "latin-1") for b in buf]
            L.H5Dvlen_reclaim(mt, space, 0, ctypes.cast(buf, ctypes.c_void_p))
            L.H5Tclose(mt)
        else:
            mt = L.H5Tcopy(ftype)
            raw = ctypes.create_string_buffer(n * size)
            if reader(mt, ctypes.cast(raw, ctypes.c_void_p)) < 0:
                L.H5Tclose(mt)
                raise IOError("HDF5: read failed")
            L.H5Tclose(mt)
            vals = [raw.raw[k * size:(k + 1) * size].split(b"\0", 1)[0].decode("utf-8" if utf8 else "latin-1") for k in range(n)]
        return np.array(vals, dtype=object).reshape(shape) if shape else vals[0]
    raise TypeError("HDF5: unsupported datatype class %d" % cls)


def _attrs(obj, system=False):
    """Attributes of an object; system=True: PyTables' own ones (CLASS, TITLE ...) instead of the data."""
    L = _lib()
    out = {}
    k = 0
    while True:
        a = L.H5Aopen_by_idx(obj, b".", 0, 0, k, 0, 0)
        if a < 0:
            break
        k += 1
        n = L.H5Aget_name(a, 0, None)
        buf = ctypes.create_string_buffer(n + 1)
        L.H5Aget_name(a, n + 1, buf)
        name = buf.value.decode()
        t, s = L.H5Aget_type(a), L.H5Aget_space(a)
        try:
            if (name in _SYSTEM_ATTRS) == system:
                out[name] = _read(a, t, s, lambda mt, p: L.H5Aread(a, mt, p))
        except TypeError:
            pass                                       # (pickled Python objects and the like: not feature data)
        finally:
            L.H5Tclose(t)
            L.H5Sclose(s)
            L.H5Aclose(a)
    return out


def _read_group(gid):
    L = _lib()
    info = _GInfo()
    if L.H5Gget_info(gid, ctypes.byref(info)) < 0:
        raise IOError("HDF5: H5Gget_info failed")
    out = {}
    for k in range(int(info.nlinks)):
        n = L.H5Lget_name_by_idx(gid, b".", 0, 0, k, None, 0, 0)
        buf = ctypes.create_string_buffer(n + 1)
        L.H5Lget_name_by_idx(gid, b".", 0, 0, k, buf, n + 1, 0)
        name = buf.value
        obj = L.H5Oopen(gid, name, 0)
        if obj < 0:
            continue
        try:
            kind = L.H5Iget_type(obj)
            if kind == H5I_GROUP:
                sub = _read_group(obj)
                title = _attrs(obj, system=True).get("TITLE") or ""
                if title.startswith(("list:", "tuple:")):             # deepdish: sequences are groups of i0, i1, ...
                    n = int(title.split(":", 1)[1])
                    seq = [sub.get("i%d" % i) for i in range(n)]
                    sub = seq if title.startswith("list:") else tuple(seq)
                elif title.startswith("nonetype:"):
                    sub = None
                out[name.decode()] = sub
            elif kind == H5I_DATASET:
                t, s = L.H5Dget_type(obj), L.H5Dget_space(obj)
                try:
                    val = _read(obj, t, s, lambda mt, p: L.H5Dread(obj, mt, 0, 0, 0, p))
                    a = _attrs(obj)
                    if "zeroarray_dtype" in a:                          # deepdish: an empty array is stored as its shape
                        val = np.zeros(tuple(int(v) for v in np.atleast_1d(val)), np.dtype(a["zeroarray_dtype"]))
                    elif a.get("strtype") in ("unicode", "ascii") and isinstance(val, np.ndarray) and val.dtype == np.uint8:
                        # deepdish: string arrays as raw bytes (UCS-4 for unicode) + the item size in characters
                        per = int(a["itemsize"]) * (4 if a["strtype"] == "unicode" else 1)
                        val = np.ascontiguousarray(val).view(("U%d" % a["itemsize"]) if a["strtype"] == "unicode" else ("S%d" % per))
                    out[name.decode()] = val
                finally:
                    L.H5Tclose(t)
                    L.H5Sclose(s)
        finally:
            L.H5Oclose(obj)
    for k, v in _attrs(gid).items():
        out.setdefault(k, v)
    return out


def read_tree(path):
    L = _lib()
    f = L.H5Fopen(os.fsencode(path), 0, 0)
    if f < 0:
        raise IOError("HDF5: cannot open %s" % path)
    try:
        root = L.H5Oopen(f, b"/", 0)
        try:
            return _read_group(root)
        finally:
            L.H5Oclose(root)
    finally:
        L.H5Fclose(f)


# ---------------------------------------------------------------------------------------------- writing
def _write_attr(obj, name, value):
    L = _lib()
    if isinstance(value, (str, bytes)):
        raw = value.encode("utf-8") if isinstance(value, str) else value
        t = L.H5Tcopy(_g("H5T_C_S1_g"))
        L.H5Tset_size(t, max(1, len(raw)))
        L.H5Tset_cset(t, 1 if isinstance(value, str) else 0)
        s = L.H5Screate(0)
        a = L.H5Acreate2(obj, name.encode(), t, s, 0, 0)
        buf = ctypes.create_string_buffer(raw, max(1, len(raw)))
        rc = L.H5Awrite(a, t, ctypes.cast(buf, ctypes.c_void_p)) if a >= 0 else -1
        L.H5Tclose(t)
    else:
        arr = np.asarray(value)
        if arr.dtype == bool:
            arr = arr.astype(np.int8)
        arr = np.require(arr, requirements="C")          # (np.ascontiguousarray would turn a scalar into shape (1,))
        t = _native(arr.dtype)
        if arr.ndim == 0:
            s = L.H5Screate(0)
        else:
            dims = (_hsize * arr.ndim)(*arr.shape)
            s = L.H5Screate_simple(arr.ndim, dims, None)
        a = L.H5Acreate2(obj, name.encode(), t, s, 0, 0)
        rc = L.H5Awrite(a, t, arr.ctypes.data_as(ctypes.c_void_p)) if a >= 0 else -1
    if a >= 0:
        L.H5Aclose(a)
    L.H5Sclose(s)
    if rc < 0:
        raise IOError("HDF5: cannot write attribute %s" % name)


def _write_dataset(gid, name, arr, compress):
    L = _lib()
    arr = np.asarray(arr)
    if arr.dtype == bool:
        arr = arr.astype(np.int8)
    arr = np.ascontiguousarray(arr)
    t = _native(arr.dtype)
    dims = (_hsize * arr.ndim)(*arr.shape)
    s = L.H5Screate_simple(arr.ndim, dims, None)
    dcpl = 0
    chunked = compress and arr.size > 300                            # deepdish: small arrays are stored plain
    if chunked:
        dcpl = L.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
        # about 1 MB per chunk, whole trailing dimensions
        row = int(np.prod(arr.shape[1:])) * arr.itemsize if arr.ndim > 1 else arr.itemsize
        lead = max(1, min(arr.shape[0], (1 << 20) // max(1, row)))
        ch = (_hsize * arr.ndim)(lead, *arr.shape[1:])
        L.H5Pset_chunk(dcpl, arr.ndim, ch)
        L.H5Pset_shuffle(dcpl)
        L.H5Pset_deflate(dcpl, int(compress))
    d = L.H5Dcreate2(gid, name.encode(), t, s, 0, dcpl, 0)
    rc = L.H5Dwrite(d, t, 0, 0, 0, arr.ctypes.data_as(ctypes.c_void_p)) if d >= 0 else -1
    if d >= 0:
        # what PyTables writes on its array nodes (readers that expect them find them)
        _write_attr(d, "CLASS", "CARRAY" if chunked else "ARRAY")
        _write_attr(d, "VERSION", "1.1" if chunked else "2.4")
        _write_attr(d, "TITLE", "")
        _write_attr(d, "FLAVOR", "numpy")
        L.H5Dclose(d)
    if dcpl:
        L.H5Pclose(dcpl)
    L.H5Sclose(s)
    if rc < 0:
        raise IOError("HDF5: cannot write dataset %s" % name)


def _write_group(gid, tree, compress):
    L = _lib()
    for k, v in tree.items():
        if isinstance(v, dict):
            g = L.H5Gcreate2(gid, k.encode(), 0, 0, 0)
            if g < 0:
                raise IOError("HDF5: cannot create group %s" % k)
            try:
                _write_attr(g, "CLASS", "GROUP")
                _write_attr(g, "VERSION", "1.0")
                _write_attr(g, "TITLE", "dict:%d" % len(v))          # how deepdish marks a dictionary level
                _write_group(g, v, compress)
            finally:
                L.H5Gclose(g)
        elif isinstance(v, (str, bytes, bool, int, float, np.generic)) or (isinstance(v, np.ndarray) and v.ndim == 0):
            _write_attr(gid, k, v[()] if isinstance(v, np.ndarray) else v)
        else:
            _write_dataset(gid, k, v, compress)


def write_tree(path, tree, compress=4):
    """tree: {name: ndarray | dict | str | number}.  compress: deflate level (0: contiguous, uncompressed)."""
    L = _lib()
    f = L.H5Fcreate(os.fsencode(path), 2, 0, 0)          # H5F_ACC_TRUNC
    if f < 0:
        raise IOError("HDF5: cannot create %s" % path)
    try:
        root = L.H5Oopen(f, b"/", 0)
        try:
            _write_attr(root, "CLASS", "GROUP")
            _write_attr(root, "PYTABLES_FORMAT_VERSION", "2.1")
            _write_attr(root, "TITLE", "")
            _write_attr(root, "VERSION", "1.0")
            _write_group(root, tree, compress)
        finally:
            L.H5Oclose(root)
    finally:
        L.H5Fclose(f)
