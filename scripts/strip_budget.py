"""Level-by-level pipeline: frame time of the stress scene against the memory budget of its arrays (smaller budget =
more, smaller strips of tiles per frame; run on a GPU box): python scripts/strip_budget.py [size]"""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd import rtx, scenes
H = rtx.hip_backend()
water, sky = H.decode_jpeg(rtx.ASSETS_DIR + '/water.jpg'), H.decode_jpeg(rtx.ASSETS_DIR + '/envmap.jpg')
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for mb in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "16384,4096,2048,1024,512,256,128".split(","))]:
    ctx = H.create(); scenes.load_stress_scene(ctx, H, water, sky); ctx.set_option(rtx.OPT_PIPELINE_BUDGET_MB, mb); ctx.commit()
    ms = [ctx.render(H.default_camera(), size, size, 4, want_stats=True)[1].kernel_ms for _ in range(4)]
    ctx.close()
    print("budget %6d MB: %.2f ms (min of %s)" % (mb, min(ms[1:]), ["%.1f" % m for m in ms]), flush=True)
