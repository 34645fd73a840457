"""CPU-only checks of the boundary: the product library builds for gfx950, loads, and exports every
symbol include/stvo_hip.h declares.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

from stvo_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    return capi.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "stvo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(stvo_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_identity_and_error_strings(lib):
    assert lib.stvo_backend_name() == b"hip-gfx950"
    assert lib.stvo_abi_version() == 3
    assert lib.stvo_error_string(0) == b"ok"
    assert b"no CPU fallback" in lib.stvo_error_string(-3)


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.stvo_ctx_create(0, 2048, 8, C.byref(h))
    assert rc == -3 and not h.value  # STVO_ERR_NO_DEVICE: never a silent CPU path
    with pytest.raises(capi.StvoError):
        capi.Context()


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "stvo-pl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".hpp", ".py", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "stvo_oracle" not in txt and "oracle_lib" not in txt and "liboracle" not in txt, os.path.join(dp, f)
