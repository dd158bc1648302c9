"""Wall clock per frame of both pipelines with frames enqueued back to back (run on a GPU box)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from raytracing_amd.rtx import ASSETS_DIR, OPT_PIPELINE, Frame, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
cam = H.default_camera()
w, h = 3840, 2160
buf = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda")
fr = Frame(w, h, 4, 0, 1)
for pipeline in (0, 1):
    ctx = H.create(); ctx.load_default_scene(water, sky); ctx.set_option(OPT_PIPELINE, pipeline); ctx.commit()
    for _ in range(3): ctx.render_device(cam, fr, buf.data_ptr())
    ctx.stats()
    n = 30
    t0 = time.perf_counter()
    for i in range(n): ctx.render_device(cam, fr, buf.data_ptr())
    t1 = time.perf_counter()
    k = ctx.stats().kernel_ms
    t2 = time.perf_counter()
    print("pipeline %d: enqueue %.3f ms/frame, to completion %.3f ms/frame, last frame by events %.3f ms" % (pipeline, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, k), flush=True)
