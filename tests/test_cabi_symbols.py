"""The C-ABI library loads and exports every symbol include/ba_amd.h declares (no GPU needed)."""

import ctypes
import os
import re

import pytest

from src import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ba_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ba_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.load_library()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ba_amd.h but not exported"
    assert set(names) == set(_native.EXPORTED_SYMBOLS)
    assert lib.ba_version() == 102


def test_struct_layouts_match_header():
    # sizeof / offsetof printed by gcc for include/ba_amd.h (x86-64 SysV)
    assert ctypes.sizeof(_native.BaOptions) == 128 and _native.BaOptions.f_scale.offset == 120
    assert ctypes.sizeof(_native.BaReport) == 3176
    assert _native.BaReport.solve_ms.offset == 80 and _native.BaReport.cost_history.offset == 104


def test_defaults():
    lib = _native.load_library()
    o = _native.BaOptions()
    lib.ba_default_options(ctypes.byref(o))
    assert o.max_iters == 2000 and o.ftol == 1e-8 and o.lambda0 == 1e-4 and o.guard_cheirality == 1


def test_no_gpu_fails_loudly():
    lib = _native.load_library()
    if lib.ba_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        _native.Solver()
