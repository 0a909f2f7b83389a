"""The C-ABI library loads and exports exactly what include/mppi_gpu_amd.h declares; without a
GPU the product path fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mppi_gpu_amd.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mppi_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_table_agree():
    from mppi_gpu_amd import _capi
    assert _declared() == sorted(_capi.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from mppi_gpu_amd import _capi
    assert os.path.exists(_capi.LIB_PATH), "run __graft_entry__.build() first"
    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], check=True,
                         capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (mppi_[a-z_0-9]+)", out))
    assert set(_declared()) <= exported
    lib = _capi.load()                       # sets prototypes for every symbol
    assert lib.mppi_version().startswith(b"mppi_gpu_amd")


def test_library_contains_gfx950_code_object():
    from mppi_gpu_amd import _capi
    data = open(_capi.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    assert b"k_rollout_fused" in data and b"k_combine" in data


def test_product_does_not_link_or_import_the_oracle():
    from mppi_gpu_amd import _capi
    out = subprocess.run(["ldd", _capi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "oracle" not in out
    pkg = os.path.join(ROOT, "mppi_gpu_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in src and "liboracle" not in src, f
                assert "orc_" not in src, f


def test_no_gpu_means_loud_failure_not_fallback():
    from mppi_gpu_amd import _capi, PointMassModel, MppiError
    lib = _capi.load()
    if lib.mppi_device_count() > 0:
        pytest.skip("a GPU is present; the no-device path cannot be exercised here")
    with pytest.raises(MppiError) as ei:
        PointMassModel(100, 50, 0.1, 2, 1)
    assert ei.value.code == -2          # MPPI_ENODEV
    assert "no CPU fallback" in str(ei.value)


def test_argument_validation_needs_no_gpu():
    from mppi_gpu_amd import _capi
    lib = _capi.load()
    h = _capi.engine_p()
    assert lib.mppi_create(10, 5, 0.1, 5, 2, 0, C.byref(h)) == -1     # S != 2A
    assert b"state_dim" in lib.mppi_last_error()
    assert lib.mppi_create(10, 5, 0.1, 10, 5, 0, C.byref(h)) == -1    # A > 4
    assert lib.mppi_create(0, 5, 0.1, 4, 2, 0, C.byref(h)) == -1
    assert lib.mppi_create(10, 5, 0.0, 4, 2, 0, C.byref(h)) == -1
    assert lib.mppi_partial_len(None) == 0
