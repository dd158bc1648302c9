"""rtx_group_*: one frame over several ranks from one process, through the C ABI.  The GPU box has one GPU, so
the ranks are several contexts on GPU 0 (a device id may be repeated): they render their row-interleaved
shares straight into the gathered buffer, or -- with RTX_GROUP_FORCE_RCCL set -- into buffers of their own
that travel to the root through RCCL send / recv pairs, the path that ranks on other GPUs take."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import make_default_context
from raytracing_amd.rtx import ASSETS_DIR, OPT_PIN_OUTPUT, OPT_STATS, Group

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _group_frame(hip, textures, n_ranks, W, H, D):
    g = Group(hip, [0] * n_ranks)
    for ctx in g.contexts:
        ctx.load_default_scene(textures["water"], textures["map.jpg"])
        ctx.set_option(OPT_STATS, 1)
    g.commit()
    frames = [g.render(hip.default_camera(), W, H, D, want_stats=True) for _ in range(3)]  # static order, cost-sorted, kept list
    g.close()
    return frames


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
def test_group_of_logical_ranks_equals_one_context(hip, textures, n_ranks):
    W, H, D = 400, 301, 4  # 301 rows: the ranks own different row counts
    single = make_default_context(hip, textures, sky="map.jpg", options=[(OPT_STATS, 1)])
    want, want_st = single.render(hip.default_camera(), W, H, D, want_stats=True)
    for got, st in _group_frame(hip, textures, n_ranks, W, H, D):
        assert np.array_equal(got, want)
        assert (st.primary_rays, st.secondary_rays, st.shadow_rays) == (want_st.primary_rays, want_st.secondary_rays, want_st.shadow_rays)


def _moved(hip, dx):
    cam = hip.default_camera()
    cam.origin[0] += dx
    return cam


@pytest.mark.parametrize("n_ranks", [1, 3])
def test_group_frames_left_on_the_root_gpu(hip, hip_runtime, textures, n_ranks):
    """rtx_group_render_device: the call enqueues and returns, the frame stays in device memory, frames alternate
    between two sets of buffers (so two may be in flight), and nothing is allocated once a size has been seen."""
    sizes = [(400, 301), (640, 360), (400, 301), (640, 360), (320, 200)]
    single = make_default_context(hip, textures, sky="map.jpg")
    want = [single.render(_moved(hip, 0.3 * k), w, h, 4) for k, (w, h) in enumerate(sizes)]
    g = Group(hip, [0] * n_ranks)
    for ctx in g.contexts:
        ctx.load_default_scene(textures["water"], textures["map.jpg"])
    g.commit()
    # two frames in flight, read back after both were enqueued
    for k in (0, 2):
        p0 = g.render_device(_moved(hip, 0.3 * k), *sizes[k], 4)
        p1 = g.render_device(_moved(hip, 0.3 * (k + 1)), *sizes[k + 1], 4)
        assert p0 != p1
        g.wait()
        assert np.array_equal(hip_runtime.to_host(p0, want[k].shape), want[k])
        assert np.array_equal(hip_runtime.to_host(p1, want[k + 1].shape), want[k + 1])
    # every size has been seen by both buffer sets now: no allocation any more, and a smaller frame fits as well
    free_before = hip_runtime.free_bytes()
    for k in (0, 1, 4, 0):
        p = g.render_device(_moved(hip, 0.3 * k), *sizes[k], 4)
        g.wait()
        assert np.array_equal(hip_runtime.to_host(p, want[k].shape), want[k])
    assert hip_runtime.free_bytes() == free_before
    g.close()


def test_group_render_into_page_locked_memory(hip, textures):
    """The host output of a group and of a context in page-locked memory of the library's (rtx_host_alloc), and in a
    buffer the library is allowed to page-lock itself (RTX_OPT_PIN_OUTPUT): the same bytes as the plain copy."""
    W, H = 400, 301
    single = make_default_context(hip, textures, sky="map.jpg")
    want = single.render(hip.default_camera(), W, H, 4)
    locked = hip.host_array((H, W, 3))
    locked[:] = 0x5A
    assert np.array_equal(single.render(hip.default_camera(), W, H, 4, out=locked), want)
    single.set_option(OPT_PIN_OUTPUT, 1)
    mine = np.full((H, W, 3), 0x5A, np.uint8)
    for _ in range(3):  # registered by the first call, found registered by the others
        mine[:] = 0x5A
        assert np.array_equal(single.render(hip.default_camera(), W, H, 4, out=mine), want)
    other = np.full((H + 7, W, 3), 0x5A, np.uint8)  # another buffer: the registration moves
    assert np.array_equal(single.render(hip.default_camera(), W, H + 7, 4, out=other)[:5], single.render(hip.default_camera(), W, H + 7, 4)[:5])
    single.set_option(OPT_PIN_OUTPUT, 0)
    g = Group(hip, [0, 0])
    for ctx in g.contexts:
        ctx.load_default_scene(textures["water"], textures["map.jpg"])
    g.commit()
    g.set_option(OPT_PIN_OUTPUT, 1)
    for out in (locked, mine, mine):
        out[:] = 0x5A
        assert np.array_equal(g.render(hip.default_camera(), W, H, 4, out=out), want)
    g.close()
    single.close()


_CHILD = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
from raytracing_amd.rtx import ASSETS_DIR, Group, hip_backend
hip = hip_backend()
tex = {{"water": hip.decode_jpeg(ASSETS_DIR + "/water.jpg"), "map.jpg": hip.decode_jpeg(ASSETS_DIR + "/map.jpg")}}
g = Group(hip, [0] * 3)
for ctx in g.contexts:
    ctx.load_default_scene(tex["water"], tex["map.jpg"])
g.commit()
for _ in range(2):
    frame = g.render(hip.default_camera(), 400, 301, 4)
# and two frames in flight, left on the GPU
import ctypes
rt = ctypes.CDLL("libamdhip64.so")
rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
ptrs = [g.render_device(hip.default_camera(), 400, 301, 4) for _ in range(2)]
g.wait()
for p in ptrs:
    got = np.zeros_like(frame)
    assert rt.hipMemcpy(got.ctypes.data, ctypes.c_void_p(p), got.size, 2) == 0
    assert np.array_equal(got, frame)
np.save({out!r}, frame)
g.close()
"""


def test_group_exchange_through_rccl(hip, textures, tmp_path):
    """RTX_GROUP_FORCE_RCCL: the two ranks that share the root's GPU send their rows to it through RCCL (send and
    receive on the one communicator of that GPU) instead of writing them in place.  In a process of its own: the
    variable is read when the group is made, and librccl is loaded there."""
    out = str(tmp_path / "frame.npy")
    env = dict(os.environ, RTX_GROUP_FORCE_RCCL="1")
    run = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT, out=out)], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    want = make_default_context(hip, textures, sky="map.jpg").render(hip.default_camera(), 400, 301, 4)
    assert np.array_equal(np.load(out), want)


def test_rt_command_line_with_two_ranks(hip, golden_small, tmp_path):
    """`rt ... -gpus N` renders through the group; with one GPU in the box only N = 1 can run here, but the
    row-interleaved path is the same code as above.  -gpus 1 must still write the reference's frame."""
    out = str(tmp_path / "rt.bmp")
    exe = os.path.join(ROOT, "raytracing_amd", "rt")
    run = subprocess.run([exe, "-out", out, "-scene", "1", "-threads", "8", "-resources", ASSETS_DIR, "-envmap",
                          os.path.join(ASSETS_DIR, "envmap.jpg"), "-gpus", "1"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    data = open(out, "rb").read()
    assert hashlib.sha256(data).hexdigest() == "144820dbbea1dca6296ae9927fdcb4de51b044b2a240c469b1283c1a7bed6ce1"  # G1
