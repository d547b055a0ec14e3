"""The C-ABI library loads and exports every symbol include/kfpos.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from roskfpos_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "kfpos.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kfpos_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(capi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.kfpos_version() == 102


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly, not compute on the CPU, when no GPU is present."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    import numpy as np
    with pytest.raises(capi.KfposError):
        capi.KfposBank(capi.MODEL_TOA, 4, np.zeros((4, 3)), init_pos=np.zeros(3))


def test_product_package_does_not_import_test_infrastructure():
    for fn in os.listdir(os.path.join(ROOT, "roskfpos_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "roskfpos_amd", fn)).read()
            assert "oracle" not in src.replace("oracle-friendly", "") or fn == "synth.py", fn
            assert "libkfpos_emu" not in src and "numpy_oracle" not in src, fn
