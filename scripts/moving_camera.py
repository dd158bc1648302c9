"""Wall-clock and kernel time per 4K frame when the camera moves every frame (run on a GPU box)."""
import os, sys, time, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from raytracing_amd.rtx import ASSETS_DIR, Frame, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
ctx = H.create(); ctx.load_default_scene(water, sky); ctx.commit()
cam = H.default_camera()
w, h = 3840, 2160
buf = torch.zeros(w * h * 3, dtype=torch.uint8, device="cuda")
fr = Frame(w, h, 4, 0, 1)
for moving in (0, 1):
    for _ in range(5): ctx.render_device(cam, fr, buf.data_ptr())
    ctx.stats()
    n = 200
    kms = []
    t0 = time.perf_counter()
    for i in range(n):
        if moving: cam.origin[0] = 0.002 * i
        ctx.render_device(cam, fr, buf.data_ptr())
        if i % 20 == 19: kms.append(ctx.stats().kernel_ms)
    ctx.stats()
    dt = (time.perf_counter() - t0) / n * 1e3
    print("%s camera: %.3f ms per frame wall clock, kernel %.3f ms (median of samples)" % ("moving" if moving else "static", dt, statistics.median(kms)), flush=True)
