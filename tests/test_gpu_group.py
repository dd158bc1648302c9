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
from raytracing_amd.rtx import ASSETS_DIR, OPT_STATS, Group

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
