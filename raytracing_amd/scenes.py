"""Synthetic scenes beyond what the reference ships: the stress scene (a million-triangle mesh and
ten thousand spheres on top of the reference's default scene), built from fixed seeds so that every
backend and every run sees the same arrays.
"""
import os

import numpy as np

from .rtx import ASSETS_DIR, MASK_ALL, MASK_MODEL, MASK_SPHERES, OPT_SPP

STRESS_SEED = 0x5EED
STRESS_SPHERES = 10000
STRESS_MATERIALS = ("glass", "mirror", "gold", "silver")


def subdivide(verts, idx, levels):
    """Each triangle into four (edge midpoints, binary32), `levels` times.  Midpoints are not shared:
    the result is a triangle soup of len(idx) * 4**levels triangles."""
    tri = np.asarray(verts, np.float32)[np.asarray(idx)]  # (n, 3, 3)
    half = np.float32(0.5)
    for _ in range(levels):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab, bc, ca = (a + b) * half, (b + c) * half, (c + a) * half
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)])
    n = len(tri)
    return np.ascontiguousarray(tri.reshape(-1, 3)), np.arange(3 * n, dtype=np.int32).reshape(n, 3)


def pcg32(seed, n, stream=1):
    """n outputs of the PCG32 generator (XSH-RR 64/32), seeded as pcg32_srandom(seed, stream) does."""
    mask = (1 << 64) - 1
    mult = 6364136223846793005
    inc = ((stream << 1) | 1) & mask
    state = 0
    state = (state * mult + inc) & mask
    state = (state + seed) & mask
    state = (state * mult + inc) & mask
    out = np.empty(n, np.uint32)
    for i in range(n):
        old = state
        state = (old * mult + inc) & mask
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        out[i] = ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF
    return out


def stress_spheres(n=STRESS_SPHERES, seed=STRESS_SEED):
    """Centres uniform in x [-60, 60], y [-3.5, 20], z [-120, -15], radii in [0.1, 0.6], one of four
    materials; five generator outputs per sphere (x, y, z, r, material), the first four as
    (u >> 8) * 2**-24.  Returns centres (n, 3) float32, radii (n,) float32, material choice (n,) in 0..3."""
    u = pcg32(seed, 5 * n).reshape(n, 5)
    f = (u[:, :4] >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    lo = np.float32([-60.0, -3.5, -120.0, 0.1])
    hi = np.float32([60.0, 20.0, -15.0, 0.6])
    v = lo + f * (hi - lo)
    return np.ascontiguousarray(v[:, :3]), np.ascontiguousarray(v[:, 3]), (u[:, 4] % np.uint32(4)).astype(np.int32)


def load_stress_scene(ctx, backend, water_rgb, sky_rgb, levels=4, n_spheres=STRESS_SPHERES, spp=4):
    """The reference's default scene with its five spheres replaced by `n_spheres` seeded ones and
    its bunny by the bunny subdivided `levels` times (4968 * 4**levels triangles), `spp` samples per
    pixel.  The context is left uncommitted."""
    ctx.load_default_scene(water_rgb, sky_rgb, mask=MASK_ALL & ~MASK_SPHERES & ~MASK_MODEL)
    centres, radii, choice = stress_spheres(n_spheres)
    mat_ids = np.array([backend.lib.rt_material_index(name.encode()) for name in STRESS_MATERIALS], np.int32)
    ctx.set_spheres(centres, radii, mat_ids[choice])
    # the reference's model: scale 45, bias (4, -7 + 1.4, -23), gold (src/Scene.cpp:153-154)
    v, f = backend.load_model(os.path.join(ASSETS_DIR, "bunny.obj"), 45.0, np.float32([4.0, np.float32(-7.0) + np.float32(1.4), -23.0]))
    verts, idx = subdivide(v, f, levels)
    ctx.set_mesh(verts, idx, backend.lib.rt_material_index(b"gold"))
    ctx.set_option(OPT_SPP, spp)
    return len(idx)
