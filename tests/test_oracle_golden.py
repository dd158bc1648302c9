"""Pins the CPU oracle to the reference: whole frames (SHA-256 of the BMP the reference build
wrote) and ray-level known answers, all produced by the reference's own functions
(tests/golden/make_golden.py).  Bit-exact, as both are x86-64 builds without FMA on the same libm.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import make_default_context
from raytracing_amd.rtx import OPT_PRUNE, OPT_STATS

FAST = ["G1", "G1a", "G1b", "G3", "G4", "G5", "S_full_d4", "S_full_d4_map", "S_full_d2", "S_planes_d1", "S_bunny_d4",
        "S_ragged"]
SLOW = ["G4k", "G5k", "G5k1", "G2", "G8"]


def _render_sha(backend, textures, cfg, tmp_path, patch=False):
    ctx = make_default_context(backend, textures, mask=cfg["mask"], sky=cfg["envmap"])
    img = ctx.render(backend.default_camera(), cfg["width"], cfg["height"], cfg["depth"])
    if patch:  # the few pixels of the largest frame that the reference decides by heap addresses (make_golden.tie_pixels)
        for row, col, b, g, r in cfg.get("address_dependent_pixels", []):
            img[row, col] = (b, g, r)
    path = str(tmp_path / "frame.bmp")
    backend.save_bmp(img, path)
    return hashlib.sha256(open(path, "rb").read()).hexdigest(), img


@pytest.mark.parametrize("tag", FAST)
def test_oracle_frame_matches_reference_sha(oracle_backend, textures, golden_frames, golden_small, tag, tmp_path):
    sha, img = _render_sha(oracle_backend, textures, golden_frames[tag], tmp_path)
    assert sha == golden_frames[tag]["bmp_sha256"]
    if tag in golden_small:
        assert np.array_equal(img, golden_small[tag])


@pytest.mark.slow
@pytest.mark.parametrize("tag", SLOW)
def test_oracle_4k_frame_matches_reference_sha(oracle_backend, textures, golden_frames, tag, tmp_path):
    if os.environ.get("RTX_SLOW") != "1":
        pytest.skip("set RTX_SLOW=1 (3840x2160 frames on the CPU)")
    sha, _ = _render_sha(oracle_backend, textures, golden_frames[tag], tmp_path, patch=True)
    assert sha == golden_frames[tag]["bmp_sha256"]


def test_oracle_own_scene_constants_match_host_layer(oracle_backend, textures, golden_frames, tmp_path):
    """The oracle's independent restatement of the scene (orx_default_scene) and the product's host
    layer (rt_scene_load_default_tex) must describe the same scene: same frame hash."""
    import ctypes as C
    lib = oracle_backend.lib
    lib.orx_default_scene.argtypes = [C.c_void_p, C.c_uint, C.c_char_p]
    ctx = oracle_backend.create()
    from raytracing_amd.rtx import ASSETS_DIR
    assert lib.orx_default_scene(ctx.handle, 31, os.path.join(ASSETS_DIR, "bunny.obj").encode()) == 0
    ctx.set_water(textures["water"], -4.5, 10.0, 5)
    ctx.set_envmap(textures["envmap.jpg"], 100)
    ctx.commit()
    cfg = golden_frames["S_full_d4"]
    img = ctx.render(oracle_backend.default_camera(), cfg["width"], cfg["height"], cfg["depth"])
    path = str(tmp_path / "f.bmp")
    oracle_backend.save_bmp(img, path)
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == cfg["bmp_sha256"]


def test_oracle_ray_kats(oracle_backend, textures, kats):
    ctx = make_default_context(oracle_backend, textures, sky="map.jpg")
    got = ctx.intersect_rays(kats["orig"], kats["dir"])
    want = kats["isect"]
    hit = want[:, 0] > 0
    assert np.array_equal(got[:, 0], want[:, 0])
    assert hit.sum() > 10000
    # hit point, normal, material of every hit: bit-exact
    assert np.array_equal(got[hit].view(np.uint32), want[hit].view(np.uint32))
    for depth, key in ((4, "rgb_d4"), (1, "rgb_d1")):
        rgb = ctx.trace_rays(kats["orig"], kats["dir"], depth)
        assert np.array_equal(rgb.view(np.uint32), kats[key].view(np.uint32)), key
    sky = ctx.trace_rays(kats["orig"], kats["dir"], 0)  # depth 0 = Sky::getColorAt
    assert np.array_equal(sky.view(np.uint32), kats["sky"].view(np.uint32))


def test_oracle_primary_directions(oracle_backend, kats):
    import ctypes as C
    from raytracing_amd.rtx import Frame
    lib = oracle_backend.lib
    cam = oracle_backend.default_camera()
    fr = Frame(80, 60, 4, 0, 1)
    out = np.zeros((60, 80, 3), np.float32)
    lib.orx_primary_dirs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.orx_primary_dirs(C.byref(cam), C.byref(fr), out.ctypes.data) == 0
    assert np.array_equal(out.view(np.uint32), kats["primary_dirs_80x60"].view(np.uint32))


def test_pruning_zero_weight_children_changes_nothing(oracle_backend, textures):
    """The HIP path skips children whose albedo weight is exactly 0; the reference traces them and
    multiplies by 0.  On the oracle both modes must give identical frames."""
    cam = oracle_backend.default_camera()
    a = make_default_context(oracle_backend, textures, sky="map.jpg").render(cam, 320, 240, 4)
    b = make_default_context(oracle_backend, textures, sky="map.jpg", options=[(OPT_PRUNE, 1)]).render(cam, 320, 240, 4)
    assert np.array_equal(a, b)


def test_oracle_ray_counts_match_survey(oracle_backend, textures):
    """sceneIntersect call counts of the default frame, as counted on an instrumented reference
    build during the survey: 2 300 642 trace + 3 965 076 shadow rays, 1 801 692 sky lookups."""
    ctx = make_default_context(oracle_backend, textures, options=[(OPT_STATS, 1)])
    _, st = ctx.render(oracle_backend.default_camera(), 800, 600, 4, want_stats=True)
    assert st.primary_rays == 480000
    assert st.primary_rays + st.secondary_rays == 2300642
    assert st.shadow_rays == 3965076
    assert st.sky_lookups == 1801692
    # and with pruning (what the GPU traces: no zero-weight children, no shadow rays towards lights that
    # can add to neither the diffuse nor the specular sum): 757 721 + 1 120 220
    ctx = make_default_context(oracle_backend, textures, options=[(OPT_STATS, 1), (OPT_PRUNE, 1)])
    _, st = ctx.render(oracle_backend.default_camera(), 800, 600, 4, want_stats=True)
    assert st.primary_rays + st.secondary_rays == 757721
    assert st.shadow_rays == 1120220
