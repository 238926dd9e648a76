"""CPU-side checks of the C ABI: the library loads and exports every symbol
that include/mvfusion.h declares (no compute calls: no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mvfusion.h")).read()
    return sorted(set(re.findall(r"\b(mvf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from mvfusion import _lib
    if not os.path.exists(_lib.SO_PATH):
        pytest.skip("libmvfusion.so not built (run __graft_entry__.build())")
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mvfusion.h but not exported"
    assert set(names) == set(_lib.EXPORTS)
    assert lib.mvf_version() == 100


def test_struct_layout():
    import ctypes as C
    from mvfusion import _lib
    assert C.sizeof(_lib.MvfView) == 128
    assert C.sizeof(_lib.MvfVisParams) == 72


def test_no_cpu_fallback():
    import torch
    from mvfusion import pooling
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        pooling.segment_csr(torch.zeros(4, 3), torch.tensor([0, 2, 4]))
