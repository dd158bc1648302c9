"""Host JPEG decoder == the reference's decoder (stb_image v2.19), byte for byte.

textures.json holds SHA-256 digests of the reference decoder's output for every JPEG the
reference ships (written by tests/golden/make_golden.py); the assets this repository carries are
checked against them, and, when the reference build is present, every file is compared directly.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
from raytracing_amd.rtx import ASSETS_DIR, RtxError

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["water.jpg", "map.jpg", "envmap.jpg"])
def test_decode_matches_reference_digest(oracle_backend, name):
    want = json.load(open(os.path.join(GOLDEN, "textures.json")))[name]
    img = oracle_backend.decode_jpeg(os.path.join(ASSETS_DIR, name))
    assert list(img.shape) == want["shape"]
    assert hashlib.sha256(img.tobytes()).hexdigest() == want["sha256"]


def test_decode_every_reference_jpeg_against_reference_decoder(oracle_backend):
    if not os.path.exists(oracle.REF_LIB_PATH) or not os.path.isdir("/root/reference/resources"):
        pytest.skip("reference build not present")
    L = C.CDLL(oracle.REF_LIB_PATH)
    L.ref_decode_jpeg.restype = C.c_void_p
    L.ref_decode_jpeg.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ref_free.argtypes = [C.c_void_p]
    files = [os.path.join("/root/reference/resources", f) for f in sorted(os.listdir("/root/reference/resources"))
             if f.lower().endswith((".jpg", ".jpeg"))] + ["/root/reference/example.jpg"]
    assert len(files) >= 8  # baseline and progressive, 4:4:4 and subsampled
    for path in files:
        w, h, n = C.c_int(), C.c_int(), C.c_int()
        p = L.ref_decode_jpeg(path.encode(), w, h, n)
        ref = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), shape=(h.value, w.value, n.value)).copy()
        L.ref_free(p)
        mine = oracle_backend.decode_jpeg(path)
        assert mine.shape == ref.shape, path
        assert np.array_equal(mine, ref), path


def test_decode_errors(oracle_backend, tmp_path):
    with pytest.raises(RtxError):
        oracle_backend.decode_jpeg(str(tmp_path / "missing.jpg"))
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"not a jpeg at all")
    with pytest.raises(RtxError):
        oracle_backend.decode_jpeg(str(bad))
    # truncated file: header parses, entropy data ends early -> still no crash
    data = open(os.path.join(ASSETS_DIR, "water.jpg"), "rb").read()
    cut = tmp_path / "cut.jpg"
    cut.write_bytes(data[: len(data) // 3])
    try:
        oracle_backend.decode_jpeg(str(cut))
    except RtxError:
        pass
