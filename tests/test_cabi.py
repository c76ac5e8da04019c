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
    assert L.acx_abi_version() == _lib.ABI_VERSION == 2


def test_default_params_match_reference_ctor():
    L = _lib.load()
    p = _lib.Serra09Params()
    L.acx_serra09_default_params(p)
    # rqa_serra09.py:31-32: oti=True, kappa=0.095, tau=1, m=9 ; essentia gammas 0.5 / 0.5
    assert (p.m, p.tau, p.oti) == (9, 1, 1) and abs(p.kappa - 0.095) < 1e-7
    assert (p.gamma_o, p.gamma_e) == (0.5, 0.5)
    assert L.acx_serra09_embed_len(2000, p) == 1991 and L.acx_serra09_embed_len(9, p) == 0


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
