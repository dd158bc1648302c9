"""The level-by-level pipeline (RTX_OPT_PIPELINE = 1): closest-hit rays, shadow rays and shading of one
recursion depth each in launches of their own over compacted ray lists, colours summed bottom-up.
It must be indistinguishable from the one-kernel pipeline through the C ABI: the same bytes, the same
ray counts -- against the oracle, against frames the reference rendered, and against RTX_OPT_PIPELINE = 0."""
import hashlib

import numpy as np
import pytest

from conftest import make_default_context
from raytracing_amd import scenes
from raytracing_amd.distributed import assemble_rows
from raytracing_amd.rtx import OPT_MESH_GRID, OPT_PIPELINE, OPT_PRUNE, OPT_SPP, OPT_STATS
from test_gpu_scenes import _random_scene

pytestmark = pytest.mark.gpu

WF = [(OPT_PIPELINE, 1)]
CASES = [  # (mask, sky, W, H, depth)
    (31, "map.jpg", 800, 600, 4),
    (31, "envmap.jpg", 400, 300, 1),
    (31, "map.jpg", 400, 300, 2),
    (17, "map.jpg", 400, 300, 1),
    (8, "map.jpg", 400, 300, 4),
    (0, "map.jpg", 200, 150, 4),
    (6, "map.jpg", 200, 150, 4),
    (31, "map.jpg", 333, 217, 3),   # ragged
    (31, "map.jpg", 17, 5, 8),      # tiny frame, maximum depth
    (31, "map.jpg", 160, 120, 0),   # depth 0: the sky only
]


@pytest.mark.parametrize("mask,sky,W,H,depth", CASES)
def test_frame_vs_oracle(hip, oracle_backend, textures, mask, sky, W, H, depth):
    cam = hip.default_camera()
    a = make_default_context(hip, textures, mask=mask, sky=sky, options=WF).render(cam, W, H, depth)
    b = make_default_context(oracle_backend, textures, mask=mask, sky=sky).render(cam, W, H, depth)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("tag", ["G1", "S_full_d4", "S_bunny_d4", "S_ragged"])
def test_frame_vs_reference_golden(hip, textures, golden_frames, golden_small, tag):
    cfg = golden_frames[tag]
    ctx = make_default_context(hip, textures, mask=cfg["mask"], sky=cfg["envmap"], options=WF)
    for _ in range(2):  # the arrays are reused by the second frame
        assert np.array_equal(ctx.render(hip.default_camera(), cfg["width"], cfg["height"], cfg["depth"]), golden_small[tag])


@pytest.mark.parametrize("prune", [1, 0])
def test_ray_counts_equal_the_one_kernel_pipeline(hip, textures, prune):
    cam = hip.default_camera()
    out = []
    for pipeline in (0, 1):
        ctx = make_default_context(hip, textures, sky="map.jpg", options=[(OPT_PIPELINE, pipeline), (OPT_STATS, 1), (OPT_PRUNE, prune)])
        f, st = ctx.render(cam, 320, 240, 4, want_stats=True)
        out.append((f, st))
    assert np.array_equal(out[0][0], out[1][0])
    a, b = out[0][1], out[1][1]
    print(b.as_dict())
    for name in ("primary_rays", "secondary_rays", "shadow_rays", "sky_lookups", "lit_lights", "mesh_box_hits", "sphere_tests", "plane_tests"):
        assert getattr(a, name) == getattr(b, name), name
    assert b.primary_rays == 320 * 240


def test_watertight_mesh_and_shares(hip, oracle_backend, textures):
    cam = hip.default_camera()
    opts = [(OPT_MESH_GRID, 0)]
    a = make_default_context(hip, textures, mask=8, sky="map.jpg", options=opts + WF).render(cam, 240, 180, 4)
    b = make_default_context(oracle_backend, textures, mask=8, sky="map.jpg", options=opts).render(cam, 240, 180, 4)
    assert np.array_equal(a, b)
    ctx = make_default_context(hip, textures, sky="map.jpg", options=WF)
    W, H, D = 301, 203, 4
    whole = ctx.render(cam, W, H, D)
    for world in (2, 3, 8):
        blocks = [ctx.render(cam, W, H, D, row_first=r, row_step=world) for r in range(world)]
        assert np.array_equal(assemble_rows(blocks, H), whole)
    assert np.array_equal(whole, make_default_context(oracle_backend, textures, sky="map.jpg").render(cam, W, H, D))


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_equal_the_oracle(hip, oracle_backend, textures, seed):
    frames = []
    for b in (oracle_backend, hip):
        ctx = b.create()
        if b is hip:
            ctx.set_option(OPT_PIPELINE, 1)
        _random_scene(ctx, b, textures, seed)
        frames.append(ctx.render(b.default_camera(), 192, 144, 5))
    assert np.array_equal(frames[0], frames[1])


def test_four_samples_and_many_spheres(hip, oracle_backend, textures):
    """The general build of the kernels: 4 samples per pixel, and the stress scene's 10 000 spheres behind
    their BVH with a subdivided bunny."""
    frames = []
    for b in (oracle_backend, hip):
        ctx = make_default_context(b, textures, options=[(OPT_SPP, 4)] + (WF if b is hip else []))
        frames.append(ctx.render(b.default_camera(), 200, 150, 4))
    assert np.array_equal(frames[0], frames[1])
    frames = []
    for b in (oracle_backend, hip):
        ctx = b.create()
        scenes.load_stress_scene(ctx, b, textures["water"], textures["map.jpg"], levels=1, n_spheres=2000)
        if b is hip:
            ctx.set_option(OPT_PIPELINE, 1)
        ctx.commit()
        frames.append(ctx.render(b.default_camera(), 96, 64, 4))
    assert np.array_equal(frames[0], frames[1])


def test_shadow_rays_drawn_one_by_one_or_in_step(hip, oracle_backend, textures):
    """RTX_OPT_RAY_POOL: with the spheres behind a tree, the lanes of a wave take their shadow rays from the list one
    by one as they finish their walks (the default), or work through 64 of them in step.  Same frame as the oracle's
    either way, same counts of rays and of lights found unhidden; also when the list ends in the middle of a wave
    (a frame of a few tiles) and with one light or none."""
    from raytracing_amd.rtx import OPT_RAY_POOL, OPT_STATS
    ref = oracle_backend.create()
    scenes.load_stress_scene(ref, oracle_backend, textures["water"], textures["map.jpg"], levels=1, n_spheres=3000)
    ref.commit()
    cam = hip.default_camera()
    for w, h, d in ((160, 120, 4), (24, 16, 3), (9, 9, 2)):
        want = ref.render(cam, w, h, d)
        stats = []
        for pool in (1, 0):
            ctx = hip.create()
            scenes.load_stress_scene(ctx, hip, textures["water"], textures["map.jpg"], levels=1, n_spheres=3000)
            ctx.set_option(OPT_PIPELINE, 1), ctx.set_option(OPT_RAY_POOL, pool), ctx.set_option(OPT_STATS, 1)
            ctx.commit()
            f, st = ctx.render(cam, w, h, d, want_stats=True)
            assert np.array_equal(f, want), (w, h, pool)
            stats.append((st.primary_rays, st.secondary_rays, st.shadow_rays, st.sky_lookups, st.lit_lights))
            ctx.set_option(OPT_STATS, 0)
            assert np.array_equal(ctx.render(cam, w, h, d), want), (w, h, pool, "no statistics")
        assert stats[0] == stats[1]
    rng = np.random.default_rng(5)
    twenty = np.concatenate([rng.uniform(-40, 40, (20, 3)) + [0, 45, -40], rng.uniform(0.05, 0.4, (20, 1))], axis=1)
    for lights, budget_mb in (([[0, 30, 10, 1.5]], 0), ([], 0), (twenty, 0), (twenty, 24), (None, 12)):
        frames = []
        for b in (oracle_backend, hip):
            ctx = b.create()
            scenes.load_stress_scene(ctx, b, textures["water"], textures["map.jpg"], levels=0, n_spheres=500)
            if lights is not None:
                ctx.set_lights(lights)  # (twenty lights: five bits of a shadow request hold the light's number)
            if b is hip:
                ctx.set_option(OPT_PIPELINE, 1)
                if budget_mb:
                    ctx.set_option(11, budget_mb)  # RTX_OPT_PIPELINE_BUDGET_MB: the frame in several strips
            ctx.commit()
            frames.append(ctx.render(b.default_camera(), 96, 72, 3))
        assert np.array_equal(frames[0], frames[1]), (None if lights is None else len(lights), budget_mb)


def test_strips_when_the_frame_exceeds_the_budget(hip, textures):
    """A frame larger than the pipeline's arrays is traced in strips of tiles; same bytes."""
    cam = hip.default_camera()
    want = make_default_context(hip, textures, sky="map.jpg").render(cam, 640, 480, 4)
    ctx = make_default_context(hip, textures, sky="map.jpg", options=WF + [(11, 64)])  # RTX_OPT_PIPELINE_BUDGET_MB
    assert np.array_equal(ctx.render(cam, 640, 480, 4), want)


@pytest.mark.parametrize("tag", ["G2", "G8"])
def test_full_size_frame_sha_equals_the_reference(hip, textures, golden_frames, tag, tmp_path):
    cfg = golden_frames[tag]
    ctx = make_default_context(hip, textures, mask=cfg["mask"], sky=cfg["envmap"], options=WF)
    img = ctx.render(hip.default_camera(), cfg["width"], cfg["height"], cfg["depth"])
    for row, col, b, g, r in cfg.get("address_dependent_pixels", []):
        img[row, col] = (b, g, r)
    path = str(tmp_path / "frame.bmp")
    hip.save_bmp(img, path)
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == cfg["bmp_sha256"]
