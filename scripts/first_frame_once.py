"""A few first frames of fresh contexts (for a kernel trace): python scripts/first_frame_once.py [mask]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd.rtx import ASSETS_DIR, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 31
for _ in range(4):
    ctx = H.create(); ctx.load_default_scene(water, sky, mask=mask); ctx.commit()
    print(ctx.render(H.default_camera(), 3840, 2160, 4, want_stats=True)[1].kernel_ms)
    ctx.close()
