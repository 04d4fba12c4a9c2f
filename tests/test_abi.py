"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/scasr.h declares (no compute calls: no GPU needed)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "scasr.h").read_text()
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from speechcatcher_amd import _abi
    if not _abi.LIB_PATH.exists():
        _abi.build()
    lib = _abi.load()
    declared = _declared_symbols()
    assert declared, "no symbols parsed from scasr.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in scasr.h but not exported"
    assert set(declared) == set(_abi.EXPORTED_SYMBOLS)
    assert lib.sc_version() >= 1


def test_argument_errors_do_not_need_a_gpu():
    from speechcatcher_amd import _abi
    lib = _abi.load()
    rc = lib.sc_gemm(None, None, 4, None, None, None, None, 4, 1, 1, 32, 0, 0, None)
    assert rc == -1
    assert b"null" in lib.sc_last_error()


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speechcatcher_amd import _abi
    from speechcatcher_amd.hip_backend import HipBackend
    with pytest.raises(_abi.ScasrError):
        HipBackend("cuda:0")
