"""Kernel time of a frame scheduled without a previous frame's costs, with the tile probe and with the projected-box
classes (run on a GPU box): the steady state of a context without schedule feedback (every frame scheduled like a first
frame, caches warm) and the very first frame of fresh contexts."""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd.rtx import ASSETS_DIR, OPT_SCHEDULE_FEEDBACK, OPT_TILE_PROBE, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
cam = H.default_camera()
def make(mask, probe, feedback=0):
    ctx = H.create(); ctx.load_default_scene(water, sky, mask=mask); ctx.set_option(OPT_SCHEDULE_FEEDBACK, feedback)
    ctx.set_option(OPT_TILE_PROBE, probe); ctx.commit()
    return ctx
WORK = (("4k", 31, (3840, 2160, 1)), ("bunny4k", 8, (3840, 2160, 1)), ("planes4k d4", 17, (3840, 2160, 1)), ("1080p", 31, (1920, 1080, 1)),
        ("4k share of 8", 31, (3840, 2160, 8)), ("8k", 31, (7680, 4320, 1)))
if len(sys.argv) > 1:
    WORK = [x for x in WORK if x[0] in sys.argv[1].split(",")]
for name, mask, (w, h, step) in WORK:
    for probe in (0, 1):
        ctx = make(mask, probe)
        ms = [ctx.render(cam, w, h, 4, row_first=0, row_step=step, want_stats=True)[1].kernel_ms for _ in range(12)]
        ctx.close()
        firsts = []
        for _ in range(4):
            c1 = make(mask, probe, 1)
            firsts.append(c1.render(cam, w, h, 4, row_first=0, row_step=step, want_stats=True)[1].kernel_ms)
            c1.close()
        print("%-14s probe %d: static order median %.3f ms, min %.3f | very first frame of a context: median %.3f, min %.3f"
              % (name, probe, statistics.median(ms[2:]), min(ms[2:]), statistics.median(firsts), min(firsts)), flush=True)
