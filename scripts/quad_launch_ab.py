"""Frame times with the quad-mode tiles inside the main launch and in a launch of their own (RTX_OPT_QUAD_KERNEL), for small
frames and for one rank's share of large ones (run on a GPU box): python scripts/quad_launch_ab.py [library]"""
import sys, os, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from conftest import make_default_context
from raytracing_amd import rtx
from raytracing_amd.rtx import ASSETS_DIR, OPT_QUAD, OPT_QUAD_KERNEL
H = rtx.Backend(os.path.join(R, "raytracing_amd", sys.argv[1] if len(sys.argv) > 1 else "librtx_hip.so"))
tex = {'water': H.decode_jpeg(ASSETS_DIR + '/water.jpg'), 'envmap.jpg': H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')}
cam = H.default_camera()
for (w, h, G) in ((800, 600, 1), (1280, 720, 1), (1920, 1080, 1), (3840, 2160, 1), (3840, 2160, 2), (3840, 2160, 4), (3840, 2160, 8), (7680, 4320, 8)):
    for quad, own in ((0, 0), (1, 0), (1, 2), (1, 1)):
        ctx = make_default_context(H, tex, options=[(OPT_QUAD, quad), (OPT_QUAD_KERNEL, own)])
        ms = [ctx.render(cam, w, h, 4, row_first=0, row_step=G, want_stats=True)[1].kernel_ms for i in range(24)]
        ctx.close()
        print('%dx%d G=%d quad=%d own=%d: median %.3f  min %.3f  max(after 4) %.3f | %s' % (w, h, G, quad, own, statistics.median(ms[4:]), min(ms[4:]), max(ms[4:]), ' '.join('%.3f' % t for t in ms[:8])), flush=True)
