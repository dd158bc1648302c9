"""Kernel time of one rank's share (rows r, r+G, ...) of the stress scene's frame for G = 1, 2, 4, 8 on one GPU
(run on a GPU box).  usage: python scripts/share_stress.py [size, default 16384]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd import scenes
from raytracing_amd.rtx import ASSETS_DIR, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
cam = H.default_camera()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = H.create()
scenes.load_stress_scene(ctx, H, water, sky)
ctx.commit()
for G in (1, 2, 4, 8):
    ts = []
    for r in sorted({0, G - 1}):
        ms = sorted(ctx.render(cam, size, size, 4, row_first=r, row_step=G, want_stats=True)[1].kernel_ms for _ in range(2 if G == 1 else 3))
        ts.append(ms[0])
    print('%d x %d, G=%d: kernel per rank %s ms' % (size, size, G, ', '.join('%.1f' % t for t in ts)), flush=True)
