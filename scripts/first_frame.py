"""Kernel time of the first frame of a view (static order) at several sizes and shares (run on a GPU box)."""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd.rtx import ASSETS_DIR, OPT_SCHEDULE_FEEDBACK, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
cam = H.default_camera()
ctx = H.create(); ctx.load_default_scene(water, sky); ctx.set_option(OPT_SCHEDULE_FEEDBACK, 0); ctx.commit()
for (w, h, step) in ((3840, 2160, 1), (1920, 1080, 1), (3840, 2160, 8), (7680, 4320, 1)):
    ms = [ctx.render(cam, w, h, 4, row_first=0, row_step=step, want_stats=True)[1].kernel_ms for _ in range(12)]
    print("%dx%d rows/%d static order: median %.3f ms, min %.3f" % (w, h, step, statistics.median(ms[2:]), min(ms[2:])), flush=True)
