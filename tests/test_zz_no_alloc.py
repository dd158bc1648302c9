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


def test_noalloc_entry_point_under_graph_capture(hip, hip_runtime, textures):
    """What the entry point is for: the call is made while its stream is being captured into a HIP graph (global
    capture mode: a hipMalloc, a hipFree or a synchronisation anywhere in the process during the capture makes it
    fail), the graph is launched twice, and the frame equals rtx_render_device's.  Free device memory does not
    change across the capture and the launches."""
    cam = hip.default_camera()
    ctx = make_default_context(hip, textures, sky="map.jpg")
    ref_ctx = make_default_context(hip, textures, sky="map.jpg")
    rt, C = hip_runtime.lib, hip_runtime.C
    for fr in (Frame(800, 600, 4, 0, 1), Frame(320, 240, 4, 0, 1)):
        want = _render(ref_ctx, cam, fr, ref_ctx.render_device)
        rows = ctx.frame_rows(fr)
        buf = torch.full((rows * fr.width * 3,), 0xA5, dtype=torch.uint8, device="cuda")
        stream, graph, graph_exec = C.c_void_p(), C.c_void_p(), C.c_void_p()
        hip_runtime.ok(rt.hipStreamCreate(C.byref(stream)), "hipStreamCreate")
        torch.cuda.synchronize()
        free_before = hip_runtime.free_bytes()
        hip_runtime.ok(rt.hipStreamBeginCapture(stream, 0), "hipStreamBeginCapture")  # 0 = hipStreamCaptureModeGlobal
        try:
            ctx.render_device_noalloc(cam, fr, buf.data_ptr(), stream.value)
        finally:
            code = rt.hipStreamEndCapture(stream, C.byref(graph))
        hip_runtime.ok(code, "hipStreamEndCapture")
        assert graph.value, "the capture was invalidated"
        assert bytes(buf[:64].cpu().numpy()) == b"\xa5" * 64  # captured, not run
        hip_runtime.ok(rt.hipGraphInstantiate(C.byref(graph_exec), graph, None, None, 0), "hipGraphInstantiate")
        free_instantiated = hip_runtime.free_bytes()
        for _ in range(2):
            buf.fill_(0xA5)
            torch.cuda.synchronize()
            hip_runtime.ok(rt.hipGraphLaunch(graph_exec, stream), "hipGraphLaunch")
            hip_runtime.ok(rt.hipStreamSynchronize(stream), "hipStreamSynchronize")
            assert np.array_equal(buf.cpu().numpy().reshape(rows, fr.width, 3), want)
        assert hip_runtime.free_bytes() == free_instantiated <= free_before
        assert ctx.stats().kernel_ms == 0.0  # a captured call records no timing events
        rt.hipGraphExecDestroy(graph_exec), rt.hipGraphDestroy(graph), rt.hipStreamDestroy(stream)


def test_noalloc_entry_point_refuses_frames_beyond_its_maximum(hip, textures):
    cam = hip.default_camera()
    ctx = make_default_context(hip, textures, sky="map.jpg")
    fr = Frame(8192, 8192, 1, 0, 1)  # 1 048 576 tiles > 524 288
    buf = torch.zeros(16, dtype=torch.uint8, device="cuda")
    with pytest.raises(RtxError) as e:
        ctx.render_device_noalloc(cam, fr, buf.data_ptr(), None)
    assert e.value.code == -5
