"""Work counters of one frame of a bench workload (run on a GPU box).  usage: python scripts/stats.py [4k] [bunny4k] ..."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from raytracing_amd import scenes
from raytracing_amd.rtx import ASSETS_DIR, OPT_STATS, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
for name in (sys.argv[1:] or ["4k"]):
    w, h, d, mask = bench.WORKLOADS[name][:4]
    ctx = H.create()
    if name.startswith("stress"):
        scenes.load_stress_scene(ctx, H, water, sky)
    else:
        ctx.load_default_scene(water, sky, mask=mask)
    ctx.set_option(OPT_STATS, 1)
    ctx.commit()
    st = ctx.render(H.default_camera(), w, h, d, want_stats=True)[1]
    print(name, {f: getattr(st, f) for f, _ in st._fields_}, flush=True)
    ctx.close()
