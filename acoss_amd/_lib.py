"""
ctypes binding of libacx.so (C ABI: include/acx.h).  This is the ONLY compute path of
the package: there is no CPU fallback.  If the shared library is missing, or no gfx950
device is present, the calls raise.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACX_LIB") or os.path.join(_HERE, "csrc", "libacx.so")   # ACX_LIB: development A/B builds

ACX_OK = 0
ACX_ERR_INVALID = -1
ACX_ERR_HIP = -2
ACX_ERR_NOMEM = -3
ACX_ERR_STATE = -4
ACX_ERR_SHORT = -5
ACX_ERR_UNSUPPORTED = -6

EXPORTS = [
    "acx_abi_version", "acx_create", "acx_destroy", "acx_last_error", "acx_set_scratch_limit",
    "acx_upload_pool", "acx_serra09_default_params", "acx_serra09_pairs", "acx_serra09_debug_pair",
    "acx_serra09_embed_len", "acx_profile_enable", "acx_profile_reset", "acx_profile_count",
    "acx_profile_get", "acx_debug_sqrt", "acx_debug_ef_sqrt", "acx_upload_pool_f64", "acx_simple_pairs",
    "acx_ef_upload_pool", "acx_earlyfusion_pairs", "acx_ef_debug_pair", "acx_sw_binary",
    "acx_chenfusion_pairs", "acx_csm_binary_sw", "acx_upload_raw_pool", "acx_download_pool",
    "acx_simple_upload_raw_pool", "acx_download_pool_f64", "acx_snf_fuse", "acx_qmax_binary",
    "acx_ef_block_features", "acx_ef_upload_raw_pool", "acx_snf_fuse_dists", "acx_grid_plan", "acx_pool_lengths", "acx_grid_run", "acx_grid_scatter", "acx_pair_grid",
    "acx_set_nonfinite_policy", "acx_nonfinite_zeroed", "acx_ef_pool_begin", "acx_ef_pool_tracks", "acx_ef_pool_end",
    "acx_set_ef_gemm", "acx_hip_versions",
    "acx_dev_alloc", "acx_dev_free", "acx_dev_read", "acx_dev_sync",
    "acx_comm_id", "acx_comm_init", "acx_comm_destroy", "acx_grid_allgather", "acx_pair_grid_ranks", "acx_set_ef_fuse",
    "acx_device_info", "acx_ef_debug_pairs",
]
ABI_VERSION = 4           # include/acx.h ACX_ABI_VERSION this shim was written against
COMM_ID_BYTES = 128

ALGO_SERRA09, ALGO_CHENFUSION, ALGO_SIMPLE, ALGO_EARLYFUSION = 0, 1, 2, 3
GRID_PLANES = {ALGO_SERRA09: 1, ALGO_CHENFUSION: 2, ALGO_SIMPLE: 1, ALGO_EARLYFUSION: 4}


class AcxError(RuntimeError):
    """A libacx call failed (HIP error, missing device, out of device memory ...)."""


class EfParams(ctypes.Structure):
    """acx_ef_params (include/acx.h); defaults = EarlyFusion ctor, earlyfusion_traile.py:44-45."""
    _fields_ = [("kappa", ctypes.c_double), ("K", ctypes.c_int32)]


class GridSpec(ctypes.Structure):
    """acx_grid_spec (include/acx.h)."""
    _fields_ = [("algo", ctypes.c_int32), ("symmetric", ctypes.c_int32), ("tile", ctypes.c_int32),
                ("world", ctypes.c_int32)]


class GridTile(ctypes.Structure):
    """acx_grid_tile (include/acx.h)."""
    _fields_ = [("row0", ctypes.c_int32), ("col0", ctypes.c_int32), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32),
                ("rank", ctypes.c_int32), ("diagonal", ctypes.c_int32), ("offset", ctypes.c_int64),
                ("cost", ctypes.c_double)]


class SimpleParams(ctypes.Structure):
    """acx_simple_params (include/acx.h); defaults = Simple ctor, simple_silva.py:26-27."""
    _fields_ = [("sslen", ctypes.c_int32), ("oti", ctypes.c_int32)]


class EfPrepParams(ctypes.Structure):
    """acx_ef_prep_params (include/acx.h); defaults = EarlyFusion ctor, earlyfusion_traile.py:44-45."""
    _fields_ = [("blocksize", ctypes.c_int32), ("mfccs_per_block", ctypes.c_int32), ("chromas_per_block", ctypes.c_int32)]


class Serra09Params(ctypes.Structure):
    """acx_serra09_params (include/acx.h); defaults = Serra09 ctor, rqa_serra09.py:31-32."""
    _fields_ = [
        ("m", ctypes.c_int32), ("tau", ctypes.c_int32), ("kappa", ctypes.c_float),
        ("oti", ctypes.c_int32), ("gamma_o", ctypes.c_float), ("gamma_e", ctypes.c_float),
        ("embed_full", ctypes.c_int32), ("pct_mode", ctypes.c_int32), ("oti_target", ctypes.c_int32),
        ("dp_start", ctypes.c_int32), ("inclusive", ctypes.c_int32), ("dmax", ctypes.c_int32), ("arith", ctypes.c_int32),
    ]


_lib = None
HIP_RUNTIME = None        # where the HIP runtime of this process came from (see _preload_hip_runtime)
HIP_VERSIONS = None       # {"build", "runtime", "runtime_from"} once the library is loaded


def _preload_hip_runtime():
    """PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  When torch and libacx live in one
    process -- the multi-GPU path hands torch device buffers to libacx -- that copy has to be the ONE runtime of
    the process whatever the import order: loaded first, libacx binds to it by soname, and a later `import torch`
    finds it already there (the other order leaves torch without a device: "No HIP GPUs are available").
    So the library itself is loaded here, by path, without importing torch (1.5 s, and single-GPU users may
    never need it).  No torch installed: the system runtime (/opt/rocm) is what libacx finds on its own."""
    global HIP_RUNTIME
    import importlib.util
    import sys
    if "torch" in sys.modules:
        HIP_RUNTIME = "torch (imported before libacx)"
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            try:
                ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
                HIP_RUNTIME = p
                return
            except OSError:
                pass
    HIP_RUNTIME = "system"


def _check_hip_version(L):
    """libacx was compiled against one HIP release and runs on whatever runtime the process loaded (torch's
    bundled one, see above).  HIP_VERSIONS records both; a different MAJOR release (no ABI promise) warns instead
    of failing in some obscure way later.  (This image: built against 7.2, PyTorch ships the 7.0 runtime.)"""
    global HIP_VERSIONS
    import warnings
    build, run = ctypes.c_int(0), ctypes.c_int(0)
    try:
        L.acx_hip_versions.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        if L.acx_hip_versions(ctypes.byref(build), ctypes.byref(run)) != 0:
            return
    except AttributeError:
        return
    # HIP_VERSION = major * 10^7 + minor * 10^5 + patch
    HIP_VERSIONS = {"build": "%d.%d" % (build.value // 10000000, build.value // 100000 % 100),
                    "runtime": "%d.%d" % (run.value // 10000000, run.value // 100000 % 100), "runtime_from": HIP_RUNTIME}
    if run.value > 0 and build.value // 10000000 != run.value // 10000000:
        warnings.warn("libacx.so was built against HIP %d.%d but runs on HIP runtime %d.%d (%s)" % (
            build.value // 10000000, build.value // 100000 % 100, run.value // 10000000, run.value // 100000 % 100, HIP_RUNTIME))


def load():
    """dlopen libacx.so and declare the prototypes.  Raises ImportError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libacx.so is not built (%s). Build it with `make -C acoss_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
    _preload_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int32)
    lp = ctypes.POINTER(ctypes.c_int64)
    pp = ctypes.POINTER(Serra09Params)
    vp = ctypes.c_void_p
    L.acx_abi_version.restype = ctypes.c_int
    if L.acx_abi_version() != ABI_VERSION:
        raise ImportError("%s has ABI version %d, this shim needs %d: rebuild it (make -C acoss_amd/csrc)"
                          % (LIB_PATH, L.acx_abi_version(), ABI_VERSION))
    L.acx_dev_alloc.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]
    L.acx_dev_free.argtypes = [vp, vp]
    L.acx_dev_read.argtypes = [vp, vp, vp, ctypes.c_int64]
    L.acx_dev_sync.argtypes = [vp]
    L.acx_comm_id.argtypes = [vp]
    L.acx_device_info.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.acx_comm_init.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32]
    L.acx_comm_destroy.argtypes = [vp]
    L.acx_grid_allgather.argtypes = [vp, vp, vp, ctypes.c_int64]
    L.acx_pair_grid_ranks.argtypes = [vp, ctypes.POINTER(GridSpec), vp, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, ctypes.c_int32]
    L.acx_create.restype = vp
    L.acx_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.acx_destroy.restype = None
    L.acx_destroy.argtypes = [vp]
    L.acx_last_error.restype = ctypes.c_char_p
    L.acx_last_error.argtypes = [vp]
    L.acx_set_scratch_limit.argtypes = [vp, ctypes.c_int64]
    L.acx_set_nonfinite_policy.argtypes = [vp, ctypes.c_int32]
    L.acx_nonfinite_zeroed.restype = ctypes.c_int64
    L.acx_nonfinite_zeroed.argtypes = [vp]
    L.acx_upload_pool.argtypes = [vp, fp, lp, ctypes.c_int32, ctypes.c_int32]
    L.acx_upload_raw_pool.argtypes = [vp, fp, lp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, lp]
    L.acx_download_pool.argtypes = [vp, fp, ctypes.c_int64]
    L.acx_serra09_default_params.restype = None
    L.acx_serra09_default_params.argtypes = [pp]
    L.acx_serra09_pairs.argtypes = [vp, ip, ctypes.c_int64, pp, fp]
    L.acx_chenfusion_pairs.argtypes = [vp, ip, ctypes.c_int64, pp, fp]
    L.acx_serra09_debug_pair.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, pp, fp, fp, fp, fp, fp, ip, fp, ip]
    L.acx_serra09_embed_len.restype = ctypes.c_int32
    L.acx_serra09_embed_len.argtypes = [ctypes.c_int32, pp]
    L.acx_profile_enable.argtypes = [vp, ctypes.c_int]
    L.acx_profile_reset.argtypes = [vp]
    L.acx_profile_count.argtypes = [vp]
    L.acx_profile_get.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_double), lp, lp]
    L.acx_debug_sqrt.argtypes = [vp, fp, ctypes.c_int64, fp]
    L.acx_debug_ef_sqrt.argtypes = [vp, fp, ctypes.c_int64, fp]
    dp = ctypes.POINTER(ctypes.c_double)
    L.acx_upload_pool_f64.argtypes = [vp, dp, lp, ctypes.c_int32, ctypes.c_int32]
    L.acx_simple_upload_raw_pool.argtypes = [vp, fp, lp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, lp]
    L.acx_download_pool_f64.argtypes = [vp, dp, ctypes.c_int64]
    L.acx_simple_pairs.argtypes = [vp, ip, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, dp]
    ep = ctypes.POINTER(EfParams)
    L.acx_ef_upload_pool.argtypes = [vp, fp, fp, fp, dp, lp, ctypes.c_int32, ip]
    L.acx_ef_pool_begin.argtypes = [vp, lp, ctypes.c_int32, ip]
    L.acx_ef_pool_tracks.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp]
    L.acx_ef_pool_end.argtypes = [vp]
    L.acx_set_ef_gemm.argtypes = [vp, ctypes.c_int32]
    L.acx_set_ef_fuse.argtypes = [vp, ctypes.c_int32]
    L.acx_earlyfusion_pairs.argtypes = [vp, ip, ctypes.c_int64, ep, fp]
    L.acx_ef_debug_pair.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ep, fp, fp, fp, ip]
    L.acx_ef_debug_pairs.argtypes = [vp, ip, ctypes.c_int64, ep, ctypes.c_int64, fp, fp, fp, ip]
    L.acx_sw_binary.argtypes = [vp, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int32, ctypes.c_int32, fp]
    L.acx_snf_fuse.argtypes = [vp, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                               ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                               ctypes.c_int32, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    L.acx_snf_fuse_dists.argtypes = [vp, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_void_p)]
    L.acx_csm_binary_sw.argtypes = [vp, fp, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, fp]
    L.acx_qmax_binary.argtypes = [vp, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int32, ctypes.c_int32, pp, fp]
    epp = ctypes.POINTER(EfPrepParams)
    L.acx_ef_block_features.argtypes = [vp, fp, ctypes.c_int64, fp, ctypes.c_int64, ctypes.c_int32, lp, ctypes.c_int32, epp,
                                        fp, fp, fp, dp]
    L.acx_ef_upload_raw_pool.argtypes = [vp, fp, lp, fp, lp, ctypes.c_int32, lp, lp, ctypes.c_int32, epp, lp]
    gp = ctypes.POINTER(GridSpec)
    L.acx_grid_plan.argtypes = [lp, ctypes.c_int32, gp, ctypes.POINTER(GridTile), ctypes.c_int64, lp, lp, dp]
    L.acx_pool_lengths.argtypes = [vp, ctypes.c_int32, lp, ctypes.c_int32, ip]
    L.acx_grid_run.argtypes = [vp, gp, vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, vp]
    L.acx_grid_scatter.argtypes = [lp, ctypes.c_int32, gp, fp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                   ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, ctypes.c_int32]
    L.acx_pair_grid.argtypes = [vp, gp, vp, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, ctypes.c_int32]
    _check_hip_version(L)
    _lib = L
    return L


ARITH = {"exact": 0, "f16x2": 1}


def serra09_params(m=9, tau=1, kappa=0.095, oti=True, gamma_o=0.5, gamma_e=0.5, embed_full=0,
                   pct_mode=0, oti_target=0, dp_start=2, inclusive=1, dmax=0, arith="exact"):
    """acx_serra09_params.  arith: "exact" (default: the f32 Gram of the arithmetic spec) or "f16x2" (opt-in, m = 9:
    two-term fp16 splits on the f16 matrix pipe -- f32-accurate, not bit-identical; include/acx.h ACX_ARITH_F16X2)."""
    return Serra09Params(int(m), int(tau), float(kappa), int(bool(oti)), float(gamma_o), float(gamma_e),
                         int(embed_full), int(pct_mode), int(oti_target), int(dp_start), int(inclusive), int(dmax),
                         int(ARITH.get(arith, arith)))


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _lptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def grid_plan(lengths, algo, symmetric, world=1, tile=0, want_tiles=False):
    """acx_grid_plan: the tile plan of the N x N pair grid -- a pure host function, no GPU needed.
    Returns dict(spec, n_tiles, floats_per_rank (world,) int64, cost_per_rank (world,) f64[, tiles])."""
    L = load()
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    spec = GridSpec(int(algo), int(bool(symmetric)), int(tile), int(world))
    nt = ctypes.c_int64(0)
    fl = np.zeros(world, np.int64)
    co = np.zeros(world, np.float64)
    rc = L.acx_grid_plan(_lptr(lengths), len(lengths), ctypes.byref(spec), None, 0, ctypes.byref(nt), _lptr(fl),
                         co.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    if rc != ACX_OK:
        raise ValueError("acx_grid_plan: bad argument")
    out = dict(spec=spec, n_tiles=int(nt.value), floats_per_rank=fl, cost_per_rank=co)
    if want_tiles:
        tiles = (GridTile * nt.value)()
        rc = L.acx_grid_plan(_lptr(lengths), len(lengths), ctypes.byref(spec), tiles, nt.value, None, None, None)
        if rc != ACX_OK:
            raise ValueError("acx_grid_plan: bad argument")
        out["tiles"] = tiles
    return out


def grid_scatter(lengths, spec, gathered, rank_stride, planes, mirror, first=0, count=-1):
    """acx_grid_scatter: gathered rank buffers (host f32) -> the (N, N) float32 planes (numpy arrays or
    memmaps, C-contiguous rows with a common leading dimension).  Pure host function."""
    L = load()
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    gathered = np.ascontiguousarray(gathered, dtype=np.float32)
    n = len(lengths)
    for P in planes:
        if P.dtype != np.float32 or P.shape != (n, n) or not P.flags["C_CONTIGUOUS"]:
            raise ValueError("grid_scatter: planes must be C-contiguous (N, N) float32")
    ptrs = (ctypes.c_void_p * len(planes))(*[P.ctypes.data for P in planes])
    rc = L.acx_grid_scatter(_lptr(lengths), n, ctypes.byref(spec), _fptr(gathered), int(rank_stride), int(first),
                            int(count), ptrs, n, int(bool(mirror)))
    if rc != ACX_OK:
        raise ValueError("acx_grid_scatter: bad argument")


def device_info(device=0):
    """{"pci_bus_id", "name", "visible_devices"} of HIP device `device` of this process (acx_device_info): what bench.py gathers per rank."""
    L = load()
    pci, name, vis = ctypes.create_string_buffer(64), ctypes.create_string_buffer(256), ctypes.c_int(0)
    rc = L.acx_device_info(int(device), pci, 64, name, 256, ctypes.byref(vis))
    if rc != 0:
        raise AcxError("acx_device_info(%d) failed (%d): no such HIP device" % (device, rc))
    return {"pci_bus_id": pci.value.decode(), "name": name.value.decode(), "visible_devices": int(vis.value)}


def comm_id():
    """acx_comm_id: a fresh RCCL communicator id (128 bytes) -- rank 0 calls it and the host hands the bytes to every rank."""
    L = load()
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    rc = L.acx_comm_id(buf)
    if rc != ACX_OK:
        msg = L.acx_last_error(None)
        raise AcxError("acx_comm_id failed (%d): %s" % (rc, msg.decode() if msg else "?"))
    return buf.raw


class DevBuf(object):
    """A device buffer owned by a libacx context (Context.dev_alloc)."""

    def __init__(self, ctx, ptr, nbytes):
        self._ctx, self.ptr, self.nbytes = ctx, ptr, nbytes

    def data_ptr(self):
        return self.ptr

    def read(self, dtype=np.float32, count=None, offset_bytes=0):
        """Copy `count` items of `dtype` starting `offset_bytes` into the buffer back to the host (drains the library's stream first)."""
        dt = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset_bytes) // dt.itemsize
        out = np.empty(int(count), dt)
        self._ctx._check(self._ctx._L.acx_dev_read(self._ctx._h, ctypes.c_void_p(out.ctypes.data),
                                                   ctypes.c_void_p(self.ptr + int(offset_bytes)), int(out.nbytes)))
        return out

    def free(self):
        if self.ptr:
            self._ctx._check(self._ctx._L.acx_dev_free(self._ctx._h, ctypes.c_void_p(self.ptr)))
            self.ptr = 0


class Context(object):
    """One libacx context = one GPU.  Single-owner (one host thread)."""

    def __init__(self, device=0, nonfinite="raise"):
        self._L = load()
        err = ctypes.c_int(0)
        self._h = self._L.acx_create(int(device), ctypes.byref(err))
        if not self._h:
            msg = self._L.acx_last_error(None)
            raise AcxError("acx_create(device=%d) failed (%d): %s"
                           % (device, err.value, msg.decode() if msg else "?"))
        self.device = int(device)
        self.n_tracks = 0
        self.lengths = None
        if nonfinite != "raise":
            self.set_nonfinite_policy(nonfinite)

    def close(self):
        if getattr(self, "_h", None):
            self._L.acx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == ACX_OK:
            return
        msg = self._L.acx_last_error(self._h)
        msg = msg.decode() if msg else "error %d" % rc
        if rc == ACX_ERR_INVALID:
            raise ValueError(msg)
        if rc == ACX_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        if rc == ACX_ERR_NOMEM:
            raise MemoryError(msg)
        # ACX_ERR_SHORT mirrors essentia's exception for inputs shorter than the stack
        raise AcxError(msg)

    def set_nonfinite_policy(self, policy):
        """'raise' (default): an upload with NaN / Inf features fails (ValueError naming the track);
        'zero': they are replaced by 0 on the device (nonfinite_zeroed() counts them)."""
        code = {"raise": 0, "reject": 0, "zero": 1}.get(policy, policy)
        self._check(self._L.acx_set_nonfinite_policy(self._h, int(code)))

    def nonfinite_zeroed(self):
        return int(self._L.acx_nonfinite_zeroed(self._h))

    def set_scratch_limit(self, nbytes):
        self._check(self._L.acx_set_scratch_limit(self._h, int(nbytes)))

    def upload_pool(self, frames, offsets):
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if frames.ndim != 2 or offsets.ndim != 1 or offsets[-1] != frames.shape[0]:
            raise ValueError("upload_pool: frames must be (sum T, dim) and offsets (n+1,) with offsets[-1] == sum T")
        self._check(self._L.acx_upload_pool(self._h, _fptr(frames),
                                            offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                            len(offsets) - 1, frames.shape[1]))
        self.n_tracks = len(offsets) - 1
        self.lengths = np.diff(offsets)

    def upload_raw_pool(self, raw, raw_offsets, fac=40):
        """Raw (sum T0, 12) f32 chroma -> block-median pooled pool on the device; returns the pooled offsets."""
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        raw_offsets = np.ascontiguousarray(raw_offsets, dtype=np.int64)
        poff = np.zeros(len(raw_offsets), np.int64)
        self._check(self._L.acx_upload_raw_pool(self._h, _fptr(raw), raw_offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                len(raw_offsets) - 1, raw.shape[1] if raw.ndim == 2 else 12, int(fac),
                                                poff.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        self.pool_offsets = poff
        return poff

    def download_pool(self, total_frames):
        out = np.empty((int(total_frames), 12), np.float32)
        self._check(self._L.acx_download_pool(self._h, _fptr(out), out.size))
        return out

    def simple_upload_raw_pool(self, raw, raw_offsets, win=200, skip=100, win_len_smooth=4):
        """Raw (sum T0, 12) f32 chroma -> SiMPle features (mean pooling, Hann smoothing, L2) on the device."""
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        raw_offsets = np.ascontiguousarray(raw_offsets, dtype=np.int64)
        poff = np.zeros(len(raw_offsets), np.int64)
        self._check(self._L.acx_simple_upload_raw_pool(self._h, _fptr(raw), raw_offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                       len(raw_offsets) - 1, raw.shape[1] if raw.ndim == 2 else 12, int(win),
                                                       int(skip), int(win_len_smooth),
                                                       poff.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        return poff

    def download_pool_f64(self, total_frames):
        out = np.empty((int(total_frames), 12), np.float64)
        self._check(self._L.acx_download_pool_f64(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.size))
        return out

    def upload_pool_f64(self, frames, offsets):
        """SiMPle features: (sum n_i, 12) f64 time-major + offsets."""
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if frames.ndim != 2 or offsets.ndim != 1 or offsets[-1] != frames.shape[0]:
            raise ValueError("upload_pool_f64: frames must be (sum n, 12) and offsets (n+1,)")
        self._check(self._L.acx_upload_pool_f64(self._h, frames.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                len(offsets) - 1, frames.shape[1]))

    def simple_pairs(self, pairs, sslen=10, oti=True):
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.empty(len(pairs), np.float64)
        self._check(self._L.acx_simple_pairs(self._h, pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                             len(pairs), int(sslen), int(bool(oti)),
                                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    def ef_upload_pool(self, tracks, slice_bytes=1 << 30):
        """tracks: list of dicts with mfccs (nb,650), ssms (nb,1225), chromas (nb,480) f32 and
        chroma_med (12,) -- the block features of EarlyFusion.load_features.  Goes up in slices of
        whole tracks of about `slice_bytes` (acx_ef_pool_begin / _tracks / _end): the host never holds a
        second copy of the collection."""
        nb = np.array([t["mfccs"].shape[0] for t in tracks], dtype=np.int64)
        dims = [int(tracks[0][k].shape[1]) for k in ("mfccs", "ssms", "chromas")] if len(tracks) else [650, 1225, 480]
        self.ef_pool_begin(nb, dims)
        row_bytes = 4 * sum(dims)
        t0 = 0
        while t0 < len(tracks):
            t1, acc = t0, 0
            while t1 < len(tracks) and (t1 == t0 or acc + int(nb[t1]) * row_bytes <= slice_bytes):
                acc += int(nb[t1]) * row_bytes
                t1 += 1
            part = tracks[t0:t1]
            mats = [np.ascontiguousarray(np.concatenate([t[k] for t in part], axis=0), dtype=np.float32)
                    for k in ("mfccs", "ssms", "chromas")]
            med = np.ascontiguousarray(np.stack([np.asarray(t["chroma_med"], dtype=np.float64) for t in part]))
            self.ef_pool_tracks(t0, t1 - t0, mats[0], mats[1], mats[2], med)
            t0 = t1
        self.ef_pool_end()

    def ef_pool_begin(self, blocks_per_track, dims=(650, 1225, 480)):
        """Start a block-feature pool of len(blocks_per_track) tracks (acx_ef_pool_begin)."""
        nb = np.ascontiguousarray(blocks_per_track, dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
        d = (ctypes.c_int32 * 3)(*[int(x) for x in dims])
        self._check(self._L.acx_ef_pool_begin(self._h, _lptr(offs), len(nb), d))
        self.ef_blocks = nb
        self._ef_dims = tuple(int(x) for x in dims)

    def ef_pool_tracks(self, first, count, mfccs, ssms, chromas, chroma_med):
        """Rows of tracks [first, first + count).  Every argument is a C-contiguous numpy array (f32 / f64 for
        the medians) OR anything with data_ptr() -- a torch tensor on this GPU is copied device to device.
        A device source must be COMPLETE when this is called (the copy is a blocking hipMemcpy that only orders
        against the NULL stream): call torch.cuda.synchronize() after the kernels that produced it.  ef_pool_end()
        fails (AcxError naming the first missing track, pool still open) unless every track was handed over."""
        def ptr(a, dtype):
            if hasattr(a, "data_ptr"):
                return ctypes.c_void_p(int(a.data_ptr())), a
            a = np.ascontiguousarray(a, dtype=dtype)
            return ctypes.c_void_p(a.ctypes.data), a
        keep = [ptr(mfccs, np.float32), ptr(ssms, np.float32), ptr(chromas, np.float32), ptr(chroma_med, np.float64)]
        self._check(self._L.acx_ef_pool_tracks(self._h, int(first), int(count), *[k[0] for k in keep]))

    def set_ef_gemm(self, mode):
        """'f16x2' (= 'default': dense rectangles of pairs, all three matrices on the matrix pipe from two fp16 terms per
        value, three MFMAs per cell), 'bf16x3' (three bf16 terms, six MFMAs: all 24 bits of every operand), 'f32',
        'bf16x3_pairwise' (one matrix at a time: mfccs / ssms with bf16x3's bits, chroma f32) or 'bf16x3_chroma_f32'
        (rectangles, chroma f32): EarlyFusion's cross-similarity GEMMs (acx_set_ef_gemm)."""
        self._check(self._L.acx_set_ef_gemm(self._h, {"bf16x3": 0, "f32": 1, "bf16x3_pairwise": 2,
                                                      "bf16x3_chroma_f32": 3, "f16x2": 4, "default": -1}.get(mode, mode)))

    def set_ef_fuse(self, mode):
        """'fast' (default: reciprocal + exp2 per kernel weight) or 'exact' (the reference's operation order with IEEE
        divisions and expf) for getWCSM's weights and the fused matrix (acx_set_ef_fuse)."""
        code = {"fast": 0, "exact": 1}.get(mode, mode)
        self._check(self._L.acx_set_ef_fuse(self._h, int(code)))

    def ef_pool_end(self):
        self._check(self._L.acx_ef_pool_end(self._h))

    def ef_block_features(self, chroma, mfcc, onsets, blocksize=20, mfccs_per_block=50, chromas_per_block=40):
        """EarlyFusion.load_features for one track on the device (acx_ef_block_features): chroma (T, 12),
        mfcc (T', ncoef) TIME-major, onsets (beat frame indices).  Returns the dict of block features."""
        chroma = np.ascontiguousarray(chroma, dtype=np.float32)
        mfcc = np.ascontiguousarray(mfcc, dtype=np.float32)
        onsets = np.ascontiguousarray(onsets, dtype=np.int64)
        ncoef = mfcc.shape[1]
        nb = max(0, len(onsets) - int(blocksize))
        out = dict(mfccs=np.zeros((nb, mfccs_per_block * ncoef), np.float32),
                   ssms=np.zeros((nb, mfccs_per_block * (mfccs_per_block - 1) // 2), np.float32),
                   chromas=np.zeros((nb, chromas_per_block * 12), np.float32), chroma_med=np.zeros(12, np.float64))
        p = EfPrepParams(int(blocksize), int(mfccs_per_block), int(chromas_per_block))
        self._check(self._L.acx_ef_block_features(
            self._h, _fptr(chroma), chroma.shape[0], _fptr(mfcc), mfcc.shape[0], ncoef, _lptr(onsets), len(onsets),
            ctypes.byref(p), _fptr(out["mfccs"]), _fptr(out["ssms"]), _fptr(out["chromas"]),
            out["chroma_med"].ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    def ef_upload_raw_pool(self, tracks, blocksize=20, mfccs_per_block=50, chromas_per_block=40):
        """tracks: list of dicts with chroma (T, 12), mfcc (T', ncoef) time-major and onsets -- the block
        features of the whole collection are built and kept on the device (acx_ef_upload_raw_pool).
        Returns the block offsets."""
        def pack(key, dtype, width=None):
            arrs = [np.ascontiguousarray(t[key], dtype=dtype) for t in tracks]
            offs = np.concatenate([[0], np.cumsum([len(a) for a in arrs])]).astype(np.int64)
            return (np.ascontiguousarray(np.concatenate(arrs, axis=0)) if len(arrs) else np.zeros(0, dtype)), offs
        ch, coff = pack("chroma", np.float32)
        mf, moff = pack("mfcc", np.float32)
        on, ooff = pack("onsets", np.int64)
        ncoef = mf.shape[1]
        boff = np.zeros(len(tracks) + 1, np.int64)
        p = EfPrepParams(int(blocksize), int(mfccs_per_block), int(chromas_per_block))
        self._check(self._L.acx_ef_upload_raw_pool(self._h, _fptr(ch), _lptr(coff), _fptr(mf), _lptr(moff), ncoef,
                                                   _lptr(on), _lptr(ooff), len(tracks), ctypes.byref(p), _lptr(boff)))
        self.ef_blocks = np.diff(boff)
        return boff

    def earlyfusion_pairs(self, pairs, kappa=0.1, K=10):
        """(n, 4) scores: mfccs, ssms, chromas, early."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.empty((len(pairs), 4), np.float32)
        p = EfParams(float(kappa), int(K))
        self._check(self._L.acx_earlyfusion_pairs(self._h, pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                  len(pairs), ctypes.byref(p), _fptr(out)))
        return out

    def ef_debug_pair(self, i, j, kappa=0.1, K=10):
        M, N = int(self.ef_blocks[i]), int(self.ef_blocks[j])
        csm = np.empty((3, M, N), np.float32)
        fused = np.empty((M, N), np.float32)
        sc = np.empty(4, np.float32)
        oti = ctypes.c_int32(0)
        p = EfParams(float(kappa), int(K))
        self._check(self._L.acx_ef_debug_pair(self._h, int(i), int(j), ctypes.byref(p), _fptr(csm), _fptr(fused),
                                              _fptr(sc), ctypes.byref(oti)))
        return dict(csm=csm, fused=fused, scores=sc, oti=int(oti.value))

    def ef_debug_pairs(self, pairs, which, kappa=0.1, K=10):
        """Intermediates of pair `which` of a list that runs as ONE batch (the multi-pair rectangles of the product path)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        if not 0 <= int(which) < len(pairs):
            raise ValueError("ef_debug_pairs: `which` = %d is not a pair of the list (%d pairs)" % (which, len(pairs)))
        i, j = (int(v) for v in pairs[which])
        if not (0 <= i < len(self.ef_blocks) and 0 <= j < len(self.ef_blocks)):
            raise ValueError("ef_debug_pairs: track index out of range in pair %d" % which)
        M, N = int(self.ef_blocks[i]), int(self.ef_blocks[j])
        csm = np.empty((3, M, N), np.float32)
        fused = np.empty((M, N), np.float32)
        sc = np.empty((len(pairs), 4), np.float32)
        oti = ctypes.c_int32(0)
        p = EfParams(float(kappa), int(K))
        self._check(self._L.acx_ef_debug_pairs(self._h, pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(pairs), ctypes.byref(p),
                                               int(which), _fptr(csm), _fptr(fused), _fptr(sc), ctypes.byref(oti)))
        return dict(csm=csm, fused=fused, scores=sc, oti=int(oti.value))

    def sw_binary(self, B):
        B = np.ascontiguousarray(B, dtype=np.uint8)
        sc = ctypes.c_float(0)
        rc = self._L.acx_sw_binary(self._h, B.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), B.shape[0], B.shape[1],
                                   ctypes.byref(sc))
        if rc == ACX_ERR_INVALID and b"Non-binary" in (self._L.acx_last_error(self._h) or b""):
            raise IOError("Non-binary elements found in input")       # alignment_tools.py:23
        self._check(rc)
        return float(sc.value)

    def csm_binary_sw(self, D, kappa=0.1):
        """smith_waterman_constrained(csm_to_binary(D, kappa)) of one f32 matrix (tests)."""
        D = np.ascontiguousarray(D, dtype=np.float32)
        sc = ctypes.c_float(0)
        self._check(self._L.acx_csm_binary_sw(self._h, _fptr(D), D.shape[0], D.shape[1], ctypes.c_double(kappa),
                                              ctypes.byref(sc)))
        return float(sc.value)

    def snf_fuse(self, Ws, Js, Vs, niters=20, reg_diag=1.0):
        """Cross-diffusion loop of similarity network fusion on the device: Ws affinity matrices (n, n) f64,
        (Js, Vs) their K-nearest-neighbour kernels as (n, K) index / weight arrays; returns the fused (n, n)."""
        m = len(Ws)
        Ws = [np.ascontiguousarray(W, dtype=np.float64) for W in Ws]
        Js = [np.ascontiguousarray(J, dtype=np.int32) for J in Js]
        Vs = [np.ascontiguousarray(V, dtype=np.float64) for V in Vs]
        n, K = Js[0].shape
        out = np.empty((n, n), np.float64)
        arr = lambda xs: (ctypes.c_void_p * m)(*[x.ctypes.data for x in xs])
        self._check(self._L.acx_snf_fuse(self._h, arr(Ws), arr(Js), arr(Vs), m, n, K, int(niters), float(reg_diag),
                                         out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    # ------------------------------------------------------------------ device buffers without torch
    def dev_alloc(self, nbytes):
        """acx_dev_alloc: a zero-filled device buffer on this context's GPU (DevBuf: .ptr, .read(dtype, count), .free())
        -- what acx_grid_run / acx_grid_allgather write into when the host holds no torch."""
        p = ctypes.c_void_p(0)
        self._check(self._L.acx_dev_alloc(self._h, int(nbytes), ctypes.byref(p)))
        return DevBuf(self, int(p.value), int(nbytes))

    def dev_sync(self):
        """hipDeviceSynchronize on this context's GPU."""
        self._check(self._L.acx_dev_sync(self._h))

    # ------------------------------------------------------------------ RCCL inside the library (no torch.distributed)
    def comm_init(self, comm_id, rank, world):
        """Join the communicator `comm_id` (bytes from comm_id(), carried to every rank by the host) as rank `rank` of
        `world`: a collective."""
        buf = ctypes.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        self._check(self._L.acx_comm_init(self._h, buf, int(rank), int(world)))

    def comm_destroy(self):
        self._check(self._L.acx_comm_destroy(self._h))

    def grid_allgather(self, local_ptr, gathered_ptr, floats_per_rank):
        """acx_grid_allgather on device pointers: rank r's floats_per_rank floats land at gathered + r * floats_per_rank."""
        self._check(self._L.acx_grid_allgather(self._h, ctypes.c_void_p(int(local_ptr)), ctypes.c_void_p(int(gathered_ptr)),
                                               int(floats_per_rank)))

    def pair_grid_ranks(self, algo, symmetric, params, planes, mirror, tile=0):
        """acx_pair_grid_ranks: the whole grid over the ranks of this context's communicator; `planes` (the (N, N) float32
        matrices) are filled on rank 0 and may be None elsewhere."""
        spec = GridSpec(int(algo), int(bool(symmetric)), int(tile), 1)
        ptrs, ld = None, 0
        if planes is not None:
            n = planes[0].shape[0]
            for P in planes:
                if P.dtype != np.float32 or P.shape != (n, n) or not P.flags["C_CONTIGUOUS"]:
                    raise ValueError("pair_grid_ranks: planes must be C-contiguous (N, N) float32")
            ptrs, ld = (ctypes.c_void_p * len(planes))(*[P.ctypes.data for P in planes]), n
        self._check(self._L.acx_pair_grid_ranks(self._h, ctypes.byref(spec), ctypes.cast(ctypes.byref(params), ctypes.c_void_p),
                                                ptrs, int(ld), int(bool(mirror))))

    # ------------------------------------------------------------------ the N x N pair grid
    def torch_device(self):
        """Where the caller allocates the tile-score buffer handed to grid_run."""
        import torch
        return torch.device("cuda", self.device)

    def pool_lengths(self, algo):
        n = ctypes.c_int32(0)
        self._check(self._L.acx_pool_lengths(self._h, int(algo), None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, np.int64)
        self._check(self._L.acx_pool_lengths(self._h, int(algo), _lptr(out), n.value, ctypes.byref(n)))
        return out

    def grid_run(self, spec, params, rank, dev_ptr, first=0, count=-1):
        """acx_grid_run: this rank's tiles [first, first + count) into the DEVICE buffer at `dev_ptr`
        (e.g. torch_tensor.data_ptr()) of floats_per_rank[rank] floats."""
        self._check(self._L.acx_grid_run(self._h, ctypes.byref(spec), ctypes.cast(ctypes.byref(params), ctypes.c_void_p),
                                         int(rank), int(first), int(count), ctypes.c_void_p(int(dev_ptr))))

    def pair_grid(self, algo, symmetric, params, planes, mirror, tile=0):
        """acx_pair_grid: the whole grid on this GPU into the (N, N) float32 planes."""
        spec = GridSpec(int(algo), int(bool(symmetric)), int(tile), 1)
        n = planes[0].shape[0]
        for P in planes:
            if P.dtype != np.float32 or P.shape != (n, n) or not P.flags["C_CONTIGUOUS"]:
                raise ValueError("pair_grid: planes must be C-contiguous (N, N) float32")
        ptrs = (ctypes.c_void_p * len(planes))(*[P.ctypes.data for P in planes])
        self._check(self._L.acx_pair_grid(self._h, ctypes.byref(spec), ctypes.cast(ctypes.byref(params), ctypes.c_void_p),
                                          ptrs, n, int(bool(mirror))))

    def snf_fuse_dists(self, Ds, K=20, niters=20, reg_diag=1.0, mu=0.5, want_ws=False):
        """doSimilarityFusion on the device (acx_snf_fuse_dists): Ds = list of (n, n) distance matrices.
        Returns (Ws or None, fused (n, n) f64)."""
        m = len(Ds)
        Ds = [np.ascontiguousarray(D, dtype=np.float64) for D in Ds]
        n = Ds[0].shape[0]
        out = np.empty((n, n), np.float64)
        Ws = [np.empty((n, n), np.float64) for _ in range(m)] if want_ws else None
        arr = lambda xs: (ctypes.c_void_p * m)(*[x.ctypes.data for x in xs])
        self._check(self._L.acx_snf_fuse_dists(self._h, arr(Ds), m, n, int(K), int(niters), float(reg_diag), float(mu),
                                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                               arr(Ws) if want_ws else None))
        return Ws, out

    def serra09_pairs(self, pairs, params=None):
        p = params or serra09_params()
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.empty(len(pairs), np.float32)
        self._check(self._L.acx_serra09_pairs(self._h, pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                              len(pairs), ctypes.byref(p), _fptr(out)))
        return out

    def chenfusion_pairs(self, pairs, params=None):
        """(K, 2) float32: column 0 Qmax, column 1 Dmax of the same cross recurrence plot."""
        p = params or serra09_params()
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.empty((len(pairs), 2), np.float32)
        self._check(self._L.acx_chenfusion_pairs(self._h, pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                 len(pairs), ctypes.byref(p), _fptr(out)))
        return out

    def serra09_embed_len(self, T, params=None):
        p = params or serra09_params()
        return int(self._L.acx_serra09_embed_len(int(T), ctypes.byref(p)))

    def serra09_debug_pair(self, i, j, params=None):
        p = params or serra09_params()
        Mq = self.serra09_embed_len(self.lengths[i], p)
        Mr = self.serra09_embed_len(self.lengths[j], p)
        if Mq <= 0 or Mr <= 0:
            Mq = Mr = 1
        d2 = np.empty((Mq, Mr), np.float32)
        eq, tq = np.empty(Mq, np.float32), np.empty(Mq, np.float32)
        er, tr = np.empty(Mr, np.float32), np.empty(Mr, np.float32)
        oti = ctypes.c_int32(0)
        score = ctypes.c_float(0)
        dims = (ctypes.c_int32 * 2)()
        self._check(self._L.acx_serra09_debug_pair(self._h, int(i), int(j), ctypes.byref(p), _fptr(d2),
                                                   _fptr(eq), _fptr(er), _fptr(tq), _fptr(tr),
                                                   ctypes.byref(oti), ctypes.byref(score), dims))
        assert (dims[0], dims[1]) == (Mq, Mr)
        return dict(d2=d2, eps_q=eq, eps_r=er, thr_q=tq, thr_r=tr, oti=int(oti.value), score=float(score.value))

    def qmax_binary(self, R, params=None):
        """Qmax / Dmax of a binary (M, N) cross recurrence plot (acx_qmax_binary)."""
        p = params or serra09_params()
        R = np.ascontiguousarray(R, dtype=np.uint8)
        score = ctypes.c_float(0)
        self._check(self._L.acx_qmax_binary(self._h, R.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                            R.shape[0], R.shape[1], ctypes.byref(p), ctypes.byref(score)))
        return float(score.value)

    def profile_enable(self, on=True):
        self._check(self._L.acx_profile_enable(self._h, int(bool(on))))

    def profile_reset(self):
        self._check(self._L.acx_profile_reset(self._h))

    def profile(self):
        out = {}
        for k in range(self._L.acx_profile_count(self._h)):
            name = ctypes.create_string_buffer(64)
            ms = ctypes.c_double(0)
            n = ctypes.c_int64(0)
            cells = ctypes.c_int64(0)
            self._check(self._L.acx_profile_get(self._h, k, name, 64, ctypes.byref(ms), ctypes.byref(n),
                                                ctypes.byref(cells)))
            out[name.value.decode()] = dict(ms=ms.value, launches=n.value, cells=cells.value)
        return out

    def debug_sqrt(self, x, ef=False):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        fn = self._L.acx_debug_ef_sqrt if ef else self._L.acx_debug_sqrt
        self._check(fn(self._h, _fptr(x), x.size, _fptr(out)))
        return out
