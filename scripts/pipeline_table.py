"""The two pipelines side by side (RTX_OPT_PIPELINE 0 = one kernel, 1 = level by level): device time per frame and
lanes per node trip of the tree walks, per workload (run on a GPU box).
usage: python scripts/pipeline_table.py [4k share8 8k share8_8k planes4k bunny4k stress4k stress]"""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd import scenes
from raytracing_amd.rtx import ASSETS_DIR, OPT_PIPELINE, OPT_STATS, hip_backend
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
cam = H.default_camera()
WORK = {  # name: (W, H, depth, mask, row_step, stress)
    "4k": (3840, 2160, 4, 31, 1, False), "8k": (7680, 4320, 4, 31, 1, False),
    "share8": (3840, 2160, 4, 31, 8, False), "share8_8k": (7680, 4320, 4, 31, 8, False),
    "planes4k": (3840, 2160, 1, 17, 1, False), "bunny4k": (3840, 2160, 4, 8, 1, False),
    "stress4k": (4096, 4096, 4, 31, 1, True), "stress": (16384, 16384, 4, 31, 1, True),
}
for name in (sys.argv[1:] or ["4k", "share8", "8k", "share8_8k", "planes4k", "bunny4k", "stress4k"]):
    w, h, d, mask, step, stress = WORK[name]
    row = []
    for pipeline in (0, 1):
        out = {}
        for stats in (0, 1):
            ctx = H.create()
            if stress:
                scenes.load_stress_scene(ctx, H, water, sky)
            else:
                ctx.load_default_scene(water, sky, mask=mask)
            ctx.set_option(OPT_PIPELINE, pipeline)
            ctx.set_option(OPT_STATS, stats)
            ctx.commit()
            if stats:
                st = ctx.render(cam, w, h, d, row_first=0, row_step=step, want_stats=True)[1]
                out["lanes"] = st.bvh_node_tests / max(st.node_wave_trips, 1)
                out["rays"] = st.primary_rays + st.secondary_rays + st.shadow_rays
            else:
                n = 3 if name == "stress" else (5 if stress else 14)
                ms = [ctx.render(cam, w, h, d, row_first=0, row_step=step, want_stats=True)[1].kernel_ms for _ in range(n)]
                out["ms"] = statistics.median(ms[2:]) if n > 3 else min(ms[1:])
            ctx.close()
        row.append(out)
    print("%-10s one kernel %9.3f ms, %4.1f lanes per node trip | level by level %9.3f ms, %4.1f lanes per node trip | %d rays" %
          (name, row[0]["ms"], row[0]["lanes"], row[1]["ms"], row[1]["lanes"], row[0]["rays"]), flush=True)
