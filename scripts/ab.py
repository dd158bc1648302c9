"""A/B timing of several builds of the library in one process (run on a GPU box): python scripts/ab.py libA.so libB.so ..."""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd import rtx
rtx._load_torch_first()
libs = [rtx.Backend(os.path.join(R, "raytracing_amd", p)) for p in sys.argv[1:]]
water, sky = libs[0].decode_jpeg(rtx.ASSETS_DIR + '/water.jpg'), libs[0].decode_jpeg(rtx.ASSETS_DIR + '/envmap.jpg')
for wl, (w, h, d, mask) in {"4k": (3840, 2160, 4, 31), "planes4k": (3840, 2160, 1, 17), "bunny4k": (3840, 2160, 4, 8)}.items():
    for rep in range(2):
        for name, B in zip(sys.argv[1:], libs):
            ctx = B.create(); ctx.load_default_scene(water, sky, mask=mask); ctx.commit()
            ms = [ctx.render(B.default_camera(), w, h, d, want_stats=True)[1].kernel_ms for _ in range(16)]
            print("%-22s %-9s median %.3f ms, min %.3f" % (name, wl, statistics.median(ms[3:]), min(ms[3:])), flush=True)
            ctx.close()
