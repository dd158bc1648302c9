"""The allocation-free entry point (rtx_render_device_noalloc) against rtx_render_device: same bytes.

Two calls in a row on one context, the second with a smaller frame, so that whatever the first call
left in the GPU's static scheduling memory (queue heads, a longer tile list) must not leak into the
second; then a third, larger one, and a frame beyond the stated maximum, which must be refused.
"""
import numpy as np
import pytest
import torch

from conftest import make_default_context
from raytracing_amd.rtx import Frame, RtxError

pytestmark = pytest.mark.gpu


def _render(ctx, cam, fr, entry):
    rows = ctx.frame_rows(fr)
    buf = torch.full((rows * fr.width * 3,), 0xA5, dtype=torch.uint8, device="cuda")  # a pattern: a missing tile shows
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    entry(cam, fr, buf.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    return buf.cpu().numpy().reshape(rows, fr.width, 3)


def test_noalloc_entry_point_matches_render_device(hip, textures):
    cam = hip.default_camera()
    ctx = make_default_context(hip, textures, sky="map.jpg")
    ref_ctx = make_default_context(hip, textures, sky="map.jpg")
    # the first is large enough for tile classes and batches, the second smaller, the third has a ragged
    # edge and is a rank's share (rows 1, 4, 7, ...)
    for fr in (Frame(800, 600, 4, 0, 1), Frame(320, 240, 4, 0, 1), Frame(1001, 700, 3, 1, 3)):
        want = _render(ref_ctx, cam, fr, ref_ctx.render_device)
        got = _render(ctx, cam, fr, ctx.render_device_noalloc)
        assert np.array_equal(got, want), "frame %dx%d differs" % (fr.width, fr.height)
        # and once more on the context that has used the ordinary entry point (its own lists stay valid)
        again = _render(ref_ctx, cam, fr, ref_ctx.render_device_noalloc)
        assert np.array_equal(again, want)
        assert np.array_equal(_render(ref_ctx, cam, fr, ref_ctx.render_device), want)


def test_noalloc_entry_point_refuses_frames_beyond_its_maximum(hip, textures):
    cam = hip.default_camera()
    ctx = make_default_context(hip, textures, sky="map.jpg")
    fr = Frame(8192, 8192, 1, 0, 1)  # 1 048 576 tiles > 524 288
    buf = torch.zeros(16, dtype=torch.uint8, device="cuda")
    with pytest.raises(RtxError) as e:
        ctx.render_device_noalloc(cam, fr, buf.data_ptr(), None)
    assert e.value.code == -5
