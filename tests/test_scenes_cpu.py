"""CPU tests of the synthetic-scene helpers and of the oracle's samples-per-pixel option (no GPU)."""
import numpy as np

from raytracing_amd import scenes
from raytracing_amd.rtx import OPT_SPP, OPT_STATS


def test_pcg32_known_answer():
    # the demo of the PCG reference implementation: pcg32_srandom(42, 54) starts with these outputs
    assert [hex(x) for x in scenes.pcg32(42, 6, 54)] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_stress_spheres_are_reproducible_and_in_range():
    c, r, m = scenes.stress_spheres(500)
    c2, r2, m2 = scenes.stress_spheres(500)
    assert np.array_equal(c, c2) and np.array_equal(r, r2) and np.array_equal(m, m2)
    assert c.dtype == np.float32 and r.dtype == np.float32
    assert (c.min(0) >= [-60, -3.5, -120]).all() and (c.max(0) <= [60, 20, -15]).all()
    assert r.min() >= 0.1 and r.max() <= 0.6
    assert set(np.unique(m)) == {0, 1, 2, 3}
    # a longer list starts with the shorter one
    c3, _, _ = scenes.stress_spheres(700)
    assert np.array_equal(c3[:500], c)


def test_subdivide_counts_and_covers_the_same_surface():
    verts = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]])
    idx = np.int32([[0, 1, 2], [1, 3, 2]])
    v, f = scenes.subdivide(verts, idx, 3)
    assert len(f) == 2 * 4 ** 3 and len(v) == 3 * len(f)
    tri = v[f]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    assert abs(area.sum() - 1.0) < 1e-5
    # orientation is kept (all normals +z)
    assert (np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])[:, 2] > 0).all()


def test_oracle_four_samples(oracle_backend, textures):
    """RTX_OPT_SPP = 4 on the oracle: four primary rays per pixel; a scene of one colour gives that colour
    (the four equal samples sum and scale exactly); the default scene's frame changes, but little."""
    flat = np.full((64, 128, 3), 77, np.uint8)
    ctx = oracle_backend.create()
    ctx.set_materials(oracle_backend.default_materials())
    ctx.set_envmap(flat)
    ctx.set_option(OPT_STATS, 1)
    ctx.commit()
    cam = oracle_backend.default_camera()
    one = ctx.render(cam, 64, 48, 2)
    ctx.set_option(OPT_SPP, 4)
    four, st = ctx.render(cam, 64, 48, 2, want_stats=True)
    assert st.primary_rays == 4 * 64 * 48
    assert np.array_equal(one, four) and (one == 77).all()

    full = oracle_backend.create()
    full.load_default_scene(textures["water"], textures["map.jpg"])
    full.commit()
    a = full.render(cam, 160, 120, 3)
    full.set_option(OPT_SPP, 4)
    b = full.render(cam, 160, 120, 3)
    diff = np.abs(a.astype(int) - b.astype(int))
    assert (a != b).any() and diff.mean() < 12.0
