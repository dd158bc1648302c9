"""Device time per frame of the two pipelines (RTX_OPT_PIPELINE 0 / 1) on several workloads (run on a GPU box).
usage: python scripts/pipelines.py [4k] [8k] [share8] [planes4k] [bunny4k] [stress4k] [small]"""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd import scenes
from raytracing_amd.rtx import ASSETS_DIR, OPT_PIPELINE, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
cam = H.default_camera()
WORK = {  # name: (W, H, depth, mask, row_step, stress)
    "small": (800, 600, 4, 31, 1, False), "1080p": (1920, 1080, 4, 31, 1, False),
    "4k": (3840, 2160, 4, 31, 1, False), "8k": (7680, 4320, 4, 31, 1, False),
    "share8": (3840, 2160, 4, 31, 8, False), "share8_8k": (7680, 4320, 4, 31, 8, False),
    "planes4k": (3840, 2160, 1, 17, 1, False), "bunny4k": (3840, 2160, 4, 8, 1, False),
    "stress4k": (4096, 4096, 4, 31, 1, True), "stress1k": (1024, 1024, 4, 31, 1, True),
}
for name in (sys.argv[1:] or ["small", "4k", "share8"]):
    w, h, d, mask, step, stress = WORK[name]
    for pipeline in (0, 1):
        ctx = H.create()
        if stress:
            scenes.load_stress_scene(ctx, H, water, sky)
        else:
            ctx.load_default_scene(water, sky, mask=mask)
        ctx.set_option(OPT_PIPELINE, pipeline)
        ctx.commit()
        n = 4 if stress else 14
        ms = [ctx.render(cam, w, h, d, row_first=0, row_step=step, want_stats=True)[1].kernel_ms for _ in range(n)]
        print("%-10s pipeline %d: median %.3f ms, min %.3f (first %.3f)" % (name, pipeline, statistics.median(ms[2:]), min(ms[2:]), ms[0]), flush=True)
        ctx.close()
