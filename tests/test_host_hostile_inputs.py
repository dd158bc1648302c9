"""The host layer's two file parsers -- the JPEG decoder and the model loader (raytracing_amd/csrc/host) -- under
AddressSanitizer and UndefinedBehaviorSanitizer with damaged inputs: every JPEG the repository ships cut short at ten
places and with bytes changed (headers, tables, entropy-coded data), model files whose header lies, whose indices point
outside the vertex array, whose numbers are not numbers.  The parsers may accept or refuse; a sanitizer report (or a
crash) fails the test.  The reference's own loaders are stb_image and src/SceneObjects.cpp:211-255 (which trusts its
header and indices, and crashes on a missing environment map, :424-444).  CPU only: `make -C oracle asan`."""
import os
import shutil
import subprocess

import pytest

from raytracing_amd.rtx import ASSETS_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZ = os.path.join(ROOT, "oracle", "_asan", "host_fuzz")
ENV = dict(os.environ, ASAN_OPTIONS="exitcode=86:detect_leaks=1:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")


@pytest.fixture(scope="module")
def fuzz():
    run = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    if run.returncode != 0 or not os.path.exists(FUZZ):
        pytest.skip("no sanitizer build here: " + run.stderr[-300:])
    return FUZZ


def _run(args, timeout=600):
    run = subprocess.run(args, env=ENV, capture_output=True, text=True, timeout=timeout)
    assert run.returncode == 0, "exit %d\n%s\n%s" % (run.returncode, run.stdout[-600:], run.stderr[-3000:])
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    return run.stdout.splitlines()


@pytest.mark.parametrize("name, cuts, flips", [("map.jpg", 10, 120), ("water.jpg", 10, 120), ("envmap.jpg", 4, 4)])
def test_damaged_jpegs(fuzz, tmp_path, name, cuts, flips):
    src = str(tmp_path / name)
    shutil.copy(os.path.join(ASSETS_DIR, name), src)
    lines = []
    for seed in (1, 2):
        lines += _run([fuzz, "jpeg", src, str(cuts), str(flips // 2), str(seed)])
    assert len(lines) == 2 * (cuts + flips // 2)
    assert all(l.startswith(("ok ", "error: ")) for l in lines), lines[:5]
    refused = sum(l.startswith("error: ") for l in lines)
    assert refused >= 2 * cuts - 2  # a file cut short is refused (a cut in the last bytes may still decode)


def test_progressive_and_other_reference_jpegs_when_present(fuzz, tmp_path):
    """The panoramas the reference ships besides the ones copied into assets/ (one of them progressive); only where the
    reference checkout is."""
    src_dir = "/root/reference/resources"
    names = [n for n in ("map1.jpg", "park.jpeg") if os.path.exists(os.path.join(src_dir, n))]
    if not names:
        pytest.skip("reference checkout not present")
    for n in names:
        src = str(tmp_path / n)
        shutil.copy(os.path.join(src_dir, n), src)
        lines = _run([fuzz, "jpeg", src, "10", "16", "7"])
        assert len(lines) == 26


def test_hostile_model_files(fuzz, tmp_path):
    good = open(os.path.join(ASSETS_DIR, "bunny.obj")).read()
    cases = {
        "empty": "",
        "header_only": "3 1\n",
        "lying_header": "100000000 100000000\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
        "huge_header": "99999999999999999999 1\nv 0 0 0\nf 1 1 1\n",
        "negative_counts": "-3 -1\nv 0 0 0\n",
        "index_zero": "3 1\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n",
        "index_beyond": "3 1\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n",
        "index_negative": "3 1\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 -2 3\n",
        "index_overflow": "3 1\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4294967297\n",
        "not_numbers": "3 1\nv a b c\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
        "huge_numbers": "3 1\nv 1e999 -1e999 nan\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
        "cut_in_a_number": good[:len(good) // 2],
        "cut_in_the_faces": good[:len(good) - 2000],
        "binary": "\x00\x01\x02\xff" * 50,
        "fewer_faces_than_promised": "3 5\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
        "good": good,
    }
    paths = []
    for name, text in cases.items():
        p = tmp_path / (name + ".obj")
        p.write_bytes(text.encode("latin-1"))
        paths.append(str(p))
    lines = _run([fuzz, "model"] + paths)
    got = dict(zip(cases, lines))
    assert got["good"].startswith("ok 2503 4968")
    assert got["huge_numbers"].startswith("ok 3 1")  # not numbers one would want, but numbers
    for name in cases:
        if name not in ("good", "huge_numbers"):
            assert got[name].startswith("error: "), (name, got[name])
