"""The tolerance build (librtx_hip_fast.so: the same sources compiled with fused multiply-add, hardware-rate
division / square root and powf on the exp / log units) against the bit-faithful build, at the tolerance the
path states: PSNR >= 50 dB and >= 99.9 % of pixels within 1/255 per channel.  The differing pixels are printed."""
import numpy as np
import pytest

from conftest import PSNR_MIN_DB, WITHIN_ONE_MIN, compare_frames, make_default_context

pytestmark = pytest.mark.gpu

CASES = [  # (mask, sky, W, H, depth)
    (31, "envmap.jpg", 800, 600, 4),   # the reference's default frame
    (31, "map.jpg", 800, 600, 2),
    (17, "envmap.jpg", 800, 600, 1),   # spheres + planes
    (8, "envmap.jpg", 800, 600, 4),    # bunny + water
    (31, "envmap.jpg", 3840, 2160, 4), # the size the metric is quoted on
]


@pytest.fixture(scope="module")
def fast():
    from raytracing_amd.rtx import hip_fast_backend
    return hip_fast_backend()


@pytest.mark.parametrize("mask,sky,W,H,depth", CASES)
def test_fast_math_frame_within_the_stated_tolerance(hip, fast, textures, mask, sky, W, H, depth):
    cam = hip.default_camera()
    exact = make_default_context(hip, textures, mask=mask, sky=sky).render(cam, W, H, depth)
    quick = make_default_context(fast, textures, mask=mask, sky=sky).render(cam, W, H, depth)
    psnr, within, bad = compare_frames(quick, exact)
    differ = int((quick != exact).any(axis=2).sum())
    print("mask %d %s %dx%d d%d: psnr %.2f dB, %.4f %% of pixels within 1/255, %d pixels differ at all, %d by more than 1: %s%s"
          % (mask, sky, W, H, depth, psnr, 100 * within, differ, len(bad), bad[:40], " ..." if len(bad) > 40 else ""))
    assert psnr >= PSNR_MIN_DB and within >= WITHIN_ONE_MIN
    # The stated tolerance is the bar; the build is far inside it because its geometry is the exact build's (only shading
    # arithmetic fuses): a handful of pixels off by one level.  A build that contracts the geometry too sits AT the bar
    # (99.89 - 99.91 % on the spheres-and-planes frame, by the compiler's whim) -- catch that before it tips over.
    assert len(bad) == 0 and differ <= W * H // 10000, (len(bad), differ)


def test_fast_math_is_a_different_build(hip, fast):
    assert fast.path != hip.path
    blob = open(fast.path, "rb").read()
    assert b"gfx950" in blob and b"k_render" in blob
