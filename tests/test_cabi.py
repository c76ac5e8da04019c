"""CPU suite, part 3: the C-ABI library loads and exports every symbol include/acx.h
declares (no compute calls without a GPU), and fails LOUDLY without a device."""
import os
import re

import pytest

from acoss_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "acx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(acx_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_library_agree():
    L = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 14
    for s in declared:
        assert hasattr(L, s), "libacx.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == declared
    assert L.acx_abi_version() == _lib.ABI_VERSION == 4


def test_default_params_match_reference_ctor():
    L = _lib.load()
    p = _lib.Serra09Params()
    L.acx_serra09_default_params(p)
    # rqa_serra09.py:31-32: oti=True, kappa=0.095, tau=1, m=9 ; essentia gammas 0.5 / 0.5
    assert (p.m, p.tau, p.oti) == (9, 1, 1) and abs(p.kappa - 0.095) < 1e-7
    assert (p.gamma_o, p.gamma_e) == (0.5, 0.5)
    # the struct ends with `arith` (ABI 2): the default is the exact f32 Gram, and the shim's layout is the header's (13 x 4 bytes)
    import ctypes
    assert p.arith == 0 and ctypes.sizeof(_lib.Serra09Params) == 52 and _lib.Serra09Params.arith.offset == 48
    assert _lib.serra09_params(arith="f16x2").arith == 1 and _lib.serra09_params().arith == 0
    assert L.acx_serra09_embed_len(2000, p) == 1991 and L.acx_serra09_embed_len(9, p) == 0


def test_missing_rccl_is_an_error_code_not_a_crash():
    """with no librccl to open, acx_comm_id must return ACX_ERR_UNSUPPORTED (the message used to be built from two
    dlerror() calls, the second of which returns NULL).  ACX_RCCL_LIB names the ONLY candidate, so a wrong path is that case."""
    import subprocess
    import sys
    code = ("import ctypes, sys\n"
            "sys.path.insert(0, %r)\n"
            "from acoss_amd import _lib\n"
            "L = _lib.load()\n"
            "buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)\n"
            "rc = L.acx_comm_id(buf)\n"
            "msg = L.acx_last_error(None)\n"
            "print('rc', rc, msg)\n"
            "sys.exit(0 if rc == -6 and b'librccl' in msg else 1)\n" % ROOT)
    env = dict(os.environ, ACX_RCCL_LIB="/nonexistent/librccl.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.AcxError):
        _lib.Context(0)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under acoss_amd/ (nor the C sources) may import,
    load or link it."""
    pkg = os.path.join(ROOT, "acoss_amd")
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if name.endswith((".py", ".hip", ".hpp", ".h")) or name == "Makefile":
                txt = open(os.path.join(dirpath, name), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), (dirpath, name)
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", txt), (dirpath, name)
                # only two files dlopen anything: _lib.py (libacx.so) and hdf5.py (the HDF5 C library) -- neither the oracle
                assert not re.search(r"(CDLL|LoadLibrary|-l\s*acx_oracle|libacx_oracle)", txt) or name in ("_lib.py", "hdf5.py"), (dirpath, name)
                if name in ("_lib.py", "hdf5.py"):
                    assert "oracle" not in txt


def _build_c_host(tmp_path):
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "c_host")
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_host.c"), "-L", lib_dir, "-lacx", "-Wl,-rpath," + lib_dir, "-lm", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_c99_and_a_c_host_links(tmp_path):
    """include/acx.h is a C header (strict C99, -pedantic -Werror) and a host written in C links libacx.so alone
    (examples/c_host.c: pool upload, acx_serra09_pairs, acx_pair_grid).  Without a GPU the program fails LOUDLY in
    acx_create (exit code 3, the library's message on stderr) -- there is no fallback to fall back to."""
    import subprocess
    import torch
    exe = _build_c_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        assert r.returncode == 0 and "covers rank first: yes" in r.stdout, (r.stdout, r.stderr)
    else:
        assert r.returncode == 3 and "acx_create failed" in r.stderr, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_c_host_on_the_gpu(tmp_path):
    """The C host's scores on the GPU: the two cover pairs rank first, and its Qmax values are what the Python shim gets
    for the same frames (the program prints them)."""
    import subprocess
    exe = _build_c_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "covers rank first: yes" in r.stdout, (r.stdout, r.stderr)
    assert r.stdout.count("Qmax") == 6
