"""Device time per frame of the bit-faithful and the tolerance build (run on a GPU box)."""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd.rtx import ASSETS_DIR, hip_backend, hip_fast_backend
for name, B in (("exact", hip_backend()), ("fast", hip_fast_backend())):
    water, sky = B.decode_jpeg(ASSETS_DIR + '/water.jpg'), B.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
    for wl, (w, h, d, mask) in {"4k": (3840, 2160, 4, 31), "planes4k": (3840, 2160, 1, 17), "bunny4k": (3840, 2160, 4, 8), "8k": (7680, 4320, 4, 31)}.items():
        ctx = B.create(); ctx.load_default_scene(water, sky, mask=mask); ctx.commit()
        ms = [ctx.render(B.default_camera(), w, h, d, want_stats=True)[1].kernel_ms for _ in range(14)]
        print("%-6s %-9s median %.3f ms, min %.3f" % (name, wl, statistics.median(ms[3:]), min(ms[3:])), flush=True)
        ctx.close()
