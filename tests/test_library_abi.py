"""The C ABI libraries load and export every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

import oracle
from raytracing_amd.rtx import HIP_LIB_PATH, HOST_SYMBOLS, RTX_SYMBOLS, Backend, RtxError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx?_[a-z0-9_]+)\s*\(", text)))


def test_binding_lists_match_headers():
    assert declared_functions("rtx.h") == sorted(RTX_SYMBOLS)
    assert declared_functions("rt_host.h") == sorted(HOST_SYMBOLS)


@pytest.mark.parametrize("path", [HIP_LIB_PATH, oracle.ORACLE_LIB_PATH], ids=["hip", "oracle"])
def test_library_exports_every_declared_symbol(path):
    assert os.path.exists(path), "%s not built" % path
    lib = ctypes.CDLL(path)
    for name in declared_functions("rtx.h") + declared_functions("rt_host.h"):
        assert hasattr(lib, name), "%s lacks %s" % (os.path.basename(path), name)


def test_hip_library_has_gfx950_code_object():
    blob = open(HIP_LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"k_render" in blob


def test_hip_backend_fails_loudly_without_a_gpu():
    """No CPU fallback: without a device rtx_create reports RTX_E_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    b = Backend(HIP_LIB_PATH)
    with pytest.raises(RtxError) as e:
        b.create()
    assert e.value.code == -3
    assert "device" in str(e.value).lower()


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(FileNotFoundError):
        Backend(str(tmp_path / "librtx_hip.so"))
