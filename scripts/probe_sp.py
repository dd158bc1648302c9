"""Measurement: distribution of the traversal stack's fill (library built with -DRTX_PROBE_SP=1): python scripts/probe_sp.py lib.so workload"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd import rtx, scenes
rtx._load_torch_first()
B = rtx.Backend(os.path.join(R, "raytracing_amd", sys.argv[1]))
water, sky = B.decode_jpeg(rtx.ASSETS_DIR + '/water.jpg'), B.decode_jpeg(rtx.ASSETS_DIR + '/envmap.jpg')
wl = sys.argv[2]
ctx = B.create()
if wl.startswith("stress"):
    scenes.load_stress_scene(ctx, B, water, sky)
    w, h = 2048, 2048
else:
    ctx.load_default_scene(water, sky, mask=31)
    w, h = 3840, 2160
ctx.commit()
ctx.render(B.default_camera(), w, h, 4, want_stats=True)
