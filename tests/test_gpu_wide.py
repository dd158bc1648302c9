"""GPU tests of the general kernel build: several samples per pixel, sphere sets behind a BVH, and the
stress scene (ten thousand spheres, a 1.27 M-triangle mesh, 16384 x 16384 pixels, 4 samples each).
The reference has none of these sizes; the oracle restates its algorithm (every sphere tested for
every ray, the 7x7x7 grid walk) and is the yardstick: frames byte for byte, hit records bit for bit."""
import numpy as np
import pytest

from conftest import make_default_context
from raytracing_amd import scenes
from raytracing_amd.rtx import OPT_PRUNE, OPT_QUAD, OPT_SPP, OPT_STATS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("quad", [1, 2])
def test_four_samples_per_pixel_equal_the_oracle(hip, oracle_backend, textures, quad):
    frames, stats = [], []
    for b in (oracle_backend, hip):
        ctx = make_default_context(b, textures, options=[(OPT_SPP, 4), (OPT_STATS, 1), (OPT_PRUNE, 1), (OPT_QUAD, quad)])
        cam = b.default_camera()
        for _ in range(1 if b is oracle_backend else 3):  # static order, cost-sorted (quad tiles), kept list
            f, st = ctx.render(cam, 200, 150, 4, want_stats=True)
            frames.append(f), stats.append(st)
    for f, st in zip(frames[1:], stats[1:]):
        assert np.array_equal(f, frames[0])
        assert st.primary_rays == 4 * 200 * 150
        assert (st.primary_rays, st.secondary_rays, st.shadow_rays, st.sky_lookups) == (
            stats[0].primary_rays, stats[0].secondary_rays, stats[0].shadow_rays, stats[0].sky_lookups)
    # and it is a different picture from the one-sample frame
    one = make_default_context(hip, textures).render(hip.default_camera(), 200, 150, 4)
    assert (one != frames[0]).mean() > 0.2


def test_sample_count_can_change_between_frames(hip, oracle_backend, textures):
    ctx = make_default_context(hip, textures)
    ref = make_default_context(oracle_backend, textures)
    cam = hip.default_camera()
    for spp in (1, 4, 4, 1, 1):
        ctx.set_option(OPT_SPP, spp), ref.set_option(OPT_SPP, spp)
        assert np.array_equal(ctx.render(cam, 96, 64, 3), ref.render(cam, 96, 64, 3))
    with pytest.raises(Exception):
        ctx.set_option(OPT_SPP, 2)


def _sphere_cloud(n, seed):
    """n random spheres in front of the default camera; among them exact duplicates with another
    material (equal distances: the first in the list must win), concentric shells, tiny and huge ones."""
    rng = np.random.default_rng(seed)
    c = np.float32([0, 2, -40]) + rng.uniform(-1, 1, (n, 3)).astype(np.float32) * np.float32([35, 10, 25])
    r = rng.uniform(0.05, 1.2, n).astype(np.float32)
    r[rng.integers(0, n, n // 10)] = 0.01           # tiny
    r[rng.integers(0, n, max(1, n // 200))] = 6.0   # a few large ones that swallow others
    m = rng.integers(0, 8, n).astype(np.int32)
    k = n // 8
    c[-k:], r[-k:] = c[:k], r[:k]                    # duplicates at the end of the list, other materials
    m[-k:] = (m[:k] + 1) % 8
    c[k:2 * k:2] = c[k + 1:2 * k + 1:2][: len(c[k:2 * k:2])]  # concentric pairs with different radii
    return c, r, m


def _rays_at(c, n, seed, far):
    rng = np.random.default_rng(seed)
    target = c[rng.integers(0, len(c), n)] + rng.normal(scale=0.3, size=(n, 3))
    o = np.float32([0, 5, 10]) + rng.normal(scale=far, size=(n, 3))
    d = target - o
    return o.astype(np.float32), np.ascontiguousarray(d / np.linalg.norm(d, axis=1, keepdims=True), np.float32)


@pytest.mark.parametrize("n", [17, 300, 6000])
def test_many_spheres_equal_the_oracle(hip, oracle_backend, textures, n):
    """More spheres than the LDS staging holds: the BVH walk finds the same sphere as the reference's
    loop over all of them, including among exactly equal distances, for near and for far-away ray
    origins (where the rounding of the sphere test reaches well beyond a small sphere's box)."""
    c, r, m = _sphere_cloud(n, n)
    out, frames = [], []
    rays = [_rays_at(c, 20000, 1, 3.0), _rays_at(c, 20000, 2, 400.0)]
    for b in (oracle_backend, hip):
        ctx = make_default_context(b, textures, mask=31 & ~1, commit=False)
        ctx.set_spheres(c, r, m)
        ctx.commit()
        out.append([ctx.intersect_rays(o, d) for o, d in rays])
        out[-1].append(ctx.trace_rays(*rays[0], depth=3))
        cam = b.default_camera()
        for _ in range(1 if b is oracle_backend else 3):
            frames.append(ctx.render(cam, 256, 160, 4))
    for a, g in zip(out[0], out[1]):
        assert (a[:, 0] > 0).sum() > 1000
        assert np.array_equal(a.view(np.uint32), g.view(np.uint32))
    for f in frames[1:]:
        assert np.array_equal(f, frames[0])


def _grazing_rays(c, r, n, seed):
    """Rays that pass a sphere's centre at about its radius, from origins 50 to 3000 units away: the distance
    from the centre is sqrt(r^2 + q * 1e-6 * L^2), q in [-1, 1.3] -- the band in which the rounding of the
    reference's sphere test (about 1.07e-6 L^2 in the squared distance) decides -- and half of the origins lie
    along a coordinate axis from their target, within a few thousandths, so that one or two direction
    components are tiny, zero or change sign."""
    rng = np.random.default_rng(seed)
    i = rng.integers(0, len(c), n)
    cc, rr = c[i].astype(np.float64), r[i].astype(np.float64)
    L = np.exp(rng.uniform(np.log(50.0), np.log(3000.0), n))
    u = rng.normal(size=(n, 3))
    axis = rng.integers(0, 3, n)
    along = rng.random(n) < 0.5
    e = np.zeros((n, 3))
    e[np.arange(n), axis] = rng.choice([-1.0, 1.0], n)
    tiny = rng.choice([0.0, 1e-4, 1.2e-3, 2.4e-3, 5e-3], (n, 3)) * rng.choice([-1.0, 1.0], (n, 3))
    tiny[np.arange(n), axis] = 0.0
    u[along] = (e + tiny)[along]
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    o = cc - u * L[:, None]                        # the ray o + t u passes through the centre ...
    v = np.cross(u, rng.normal(size=(n, 3)))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    q = rng.uniform(-1.0, 1.3, n)
    d2 = np.maximum(rr * rr + q * 1e-6 * L * L, 0.0)
    target = cc + v * np.sqrt(d2)[:, None]         # ... and this one passes it at sqrt(d2)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)


@pytest.mark.parametrize("n", [40, 3000])
def test_grazing_rays_from_far_origins_equal_the_oracle(hip, oracle_backend, textures, n):
    """What the sphere tree's walk may cull depends on the rounding of the reference's sphere test, which grows
    with the distance to the sphere: rays from far away that graze small spheres, many of them almost parallel
    to an axis, hit exactly what the loop over all spheres hits."""
    c, r, m = _sphere_cloud(n, 7 * n)
    r[::3] = 0.02
    rays = [_grazing_rays(c, r, 30000, 11), _grazing_rays(c, r, 30000, 12)]
    out = []
    for b in (oracle_backend, hip):
        ctx = make_default_context(b, textures, mask=31 & ~1, commit=False)
        ctx.set_spheres(c, r, m)
        ctx.commit()
        out.append([ctx.intersect_rays(o, d) for o, d in rays])
    for a, g in zip(out[0], out[1]):
        assert (a[:, 0] > 0).sum() > 1000
        assert np.array_equal(a.view(np.uint32), g.view(np.uint32))


def test_sphere_list_order_decides_ties(hip, oracle_backend, textures):
    """Twenty copies of one sphere, each with its own material: the first one in the list is the one hit."""
    mats = np.tile(hip.default_materials()[:1], (20, 1))
    mats[:, 5] = np.linspace(0.05, 1.0, 20)
    frames = []
    for b in (oracle_backend, hip):
        ctx = b.create()
        ctx.set_materials(mats)
        ctx.set_lights([[0, 30, 10, 1.5]])
        order = np.arange(20)[::-1].copy()
        ctx.set_spheres(np.tile(np.float32([[0, 0, -12]]), (20, 1)), np.full(20, 3.0), order)
        ctx.set_envmap(textures["map.jpg"])
        ctx.commit()
        frames.append(ctx.render(b.default_camera(), 128, 96, 2))
    assert np.array_equal(frames[0], frames[1])


@pytest.mark.parametrize("size, levels, n_spheres, rows", [(2048, 2, 2000, (300, 700, 1000, 1300, 1480, 1800)),
                                                           (16384, 4, scenes.STRESS_SPHERES, (7000, 9000))])
def test_stress_scene_rows_equal_the_oracle(hip, oracle_backend, textures, size, levels, n_spheres, rows):
    """The stress configuration -- the default scene with ten thousand seeded spheres and the bunny
    subdivided to 1 271 808 triangles, 16384 x 16384 pixels, 4 samples per pixel -- rendered whole on
    the GPU; the oracle renders a few full-width rows of it (through the sphere cloud, through the
    top of the mesh; the smaller edition also through its widest part) and those rows are equal byte
    for byte.  A smaller edition of the same scene runs first.  The whole frame is then rendered again
    as two row-interleaved halves, as two ranks would: the same frame."""
    hip_ctx = hip.create()
    nt = scenes.load_stress_scene(hip_ctx, hip, textures["water"], textures["envmap.jpg"], levels, n_spheres)
    assert nt == 4968 * 4 ** levels
    hip_ctx.commit()
    cam = hip.default_camera()
    frame = hip_ctx.render(cam, size, size, 4)
    ref = oracle_backend.create()
    scenes.load_stress_scene(ref, oracle_backend, textures["water"], textures["envmap.jpg"], levels, n_spheres)
    ref.commit()
    seen = []
    for row in rows:
        want = ref.render(cam, size, size, 4, row_first=row, row_step=size)
        assert want.shape == (1, size, 3)
        assert np.array_equal(frame[row], want[0]), row
        seen.append(want[0])
    assert len(np.unique(np.concatenate(seen), axis=0)) > 2000  # not trivial rows
    for first in (0, 1):
        half = hip_ctx.render(cam, size, size, 4, row_first=first, row_step=2)
        assert np.array_equal(half, frame[first::2])
    # a second and third frame of the same view (cost-sorted order, quad tiles; kept list)
    for _ in range(2):
        assert np.array_equal(hip_ctx.render(cam, size, size, 4), frame)


@pytest.mark.parametrize("quad", [1, 2])
def test_every_frame_writes_every_pixel_with_four_samples(hip, oracle_backend, textures, quad):
    """The same check as for one sample per pixel: a pattern in the frame buffer before every render, so
    that a tile missing from the list (static classes, cost bins, kept list) cannot hide behind an old
    frame.  Pooled sample paths in the plain tiles, samples in turn in the quad-mode quarters."""
    import torch
    from raytracing_amd.rtx import Frame
    w, h = 203, 131
    ctx = make_default_context(hip, textures, sky="map.jpg", options=[(OPT_SPP, 4), (OPT_QUAD, quad)])
    want = make_default_context(oracle_backend, textures, sky="map.jpg", options=[(OPT_SPP, 4)]).render(oracle_backend.default_camera(), w, h, 3)
    cam = hip.default_camera()
    buf = torch.empty(h * w * 3, dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.Stream()
    for _ in range(3):
        buf.fill_(0x5A)
        torch.cuda.synchronize()
        ctx.render_device(cam, Frame(w, h, 3, 0, 1), buf.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy().reshape(h, w, 3), want)
