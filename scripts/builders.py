"""Frame time and commit time with each mesh tree builder (RTX_OPT_BVH_BUILDER 1 Morton GPU / 2 SAH host / 3 PLOC GPU), on a GPU box."""
import os, sys, time, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from raytracing_amd.rtx import ASSETS_DIR, OPT_BVH_BUILDER, OPT_STATS, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
cam = H.default_camera()
ref = None
for builder in [int(b) for b in (sys.argv[1:] or [2, 0, 1, 3])]:
    for mask, name in ((31, "4k"), (8, "bunny4k")):
        ctx = H.create(); ctx.load_default_scene(water, sky, mask=mask); ctx.set_option(OPT_BVH_BUILDER, builder)
        t0 = time.perf_counter(); ctx.commit(); dt = time.perf_counter() - t0
        ms = [ctx.render(cam, 3840, 2160, 4, want_stats=True)[1].kernel_ms for _ in range(14)]
        ctx.close()
        ctx = H.create(); ctx.load_default_scene(water, sky, mask=mask); ctx.set_option(OPT_BVH_BUILDER, builder); ctx.set_option(OPT_STATS, 1); ctx.commit()
        img, st = ctx.render(cam, 1920, 1080, 4, want_stats=True)
        if ref is None: ref = {}
        same = ref.setdefault(name, img) is img or np.array_equal(ref[name], img)
        print("builder %d %-8s commit %.1f ms, frame median %.3f ms min %.3f; 1080p node tests %d, same frame %s" % (builder, name, dt * 1e3, statistics.median(ms[3:]), min(ms[3:]), st.bvh_node_tests, same), flush=True)
        ctx.close()
