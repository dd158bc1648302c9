"""Kernel time of one rank's share of a frame (rows r, r+G, ...) for G = 1, 2, 4, 8 on one GPU (run on a GPU box).
usage: python scripts/share.py [quad mode 0..2, default 1] [width height] [quad-mode tiles in a kernel of their own: 0/1, default 1]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from conftest import make_default_context
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_QUAD, OPT_QUAD_KERNEL
H = hip_backend()
tex = {'water': H.decode_jpeg(ASSETS_DIR + '/water.jpg'), 'envmap.jpg': H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')}
cam = H.default_camera()
quad = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, Hh = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
own = int(sys.argv[4]) if len(sys.argv) > 4 else 1
D = 4
print("RTX_OPT_QUAD = %d, RTX_OPT_QUAD_KERNEL = %d, %d x %d" % (quad, own, W, Hh))
for G in (1, 2, 4, 8):
    ts = []
    for r in range(min(G, 3)):
        ctx = make_default_context(H, tex, options=[(OPT_QUAD, quad), (OPT_QUAD_KERNEL, own)])
        ms = sorted(ctx.render(cam, W, Hh, D, row_first=r, row_step=G, want_stats=True)[1].kernel_ms for i in range(10))
        ts.append(ms[2])  # (the first frames of a view are scheduled without costs)
        ctx.close()
    print('G=%d: kernel per rank %s ms' % (G, ', '.join('%.3f' % t for t in ts)), flush=True)
