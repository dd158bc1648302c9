"""A few 4K frames through the level-by-level pipeline; RTX_WF_TIMING=1 prints per-launch times (run on a GPU box)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd.rtx import ASSETS_DIR, OPT_PIPELINE, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
ctx = H.create(); ctx.load_default_scene(water, sky); ctx.set_option(OPT_PIPELINE, 1); ctx.commit()
for i in range(4):
    _, st = ctx.render(H.default_camera(), 3840, 2160, 4, want_stats=True)
    print("frame", i, "kernel_ms %.3f" % st.kernel_ms, file=sys.stderr, flush=True)
