"""Kernel time of one rank's share of a frame (rows r, r+G, ...) for G = 1, 2, 4, 8 on one GPU (run on a GPU box).
usage: python scripts/share.py [quad mode 0..2, default 1] [width height]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_QUAD
H = hip_backend()
tex = {'water': H.decode_jpeg(ASSETS_DIR + '/water.jpg'), 'envmap.jpg': H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')}
cam = H.default_camera()
quad = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, Hh = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
D = 4
print("RTX_OPT_QUAD = %d, %d x %d" % (quad, W, Hh))
for G in (1, 2, 4, 8):
    ts = []
    for r in range(min(G, 3)):
        ctx = make_default_context(H, tex, options=[(OPT_QUAD, quad)])
        ms = sorted(ctx.render(cam, W, Hh, D, row_first=r, row_step=G, want_stats=True)[1].kernel_ms for i in range(10))
        ts.append(ms[2])  # (the first frames of a view are scheduled without costs)
        ctx.close()
    print('G=%d: kernel per rank %s ms' % (G, ', '.join('%.3f' % t for t in ts)), flush=True)
