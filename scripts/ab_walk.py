"""A/B of several builds of the library on the bench workloads, with the walks' statistics (run on a GPU box):
    python scripts/ab_walk.py [--workloads 4k,bunny4k,...] libA.so libB.so ...
Per build and workload: kernel time (median of the frames after the third), node tests per ray, lanes per node trip."""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd import rtx, scenes
rtx._load_torch_first()
args = sys.argv[1:]
names = "4k,planes4k,bunny4k,8k,stress4k"
if args and args[0] == "--workloads":
    names, args = args[1], args[2:]
libs = [rtx.Backend(os.path.join(R, "raytracing_amd", p)) for p in args]
water, sky = libs[0].decode_jpeg(rtx.ASSETS_DIR + '/water.jpg'), libs[0].decode_jpeg(rtx.ASSETS_DIR + '/envmap.jpg')
WL = {"4k": (3840, 2160, 4, 31), "planes4k": (3840, 2160, 1, 17), "bunny4k": (3840, 2160, 4, 8), "8k": (7680, 4320, 4, 31),
      "stress4k": (4096, 4096, 4, 31), "stress": (16384, 16384, 4, 31)}
for wl in names.split(","):
    w, h, d, mask = WL[wl]
    stress = wl.startswith("stress")
    for name, B in zip(args, libs):
        def make(stats):
            ctx = B.create()
            if stress:
                scenes.load_stress_scene(ctx, B, water, sky)
            else:
                ctx.load_default_scene(water, sky, mask=mask)
            if stats:
                ctx.set_option(rtx.OPT_STATS, 1)
            for pair in filter(None, os.environ.get("RTX_AB_OPTS", "").split(",")):  # id=value pairs of rtx_set_option
                ctx.set_option(int(pair.split("=")[0]), int(pair.split("=")[1]))
            ctx.commit()
            return ctx
        ctx = make(False)
        n = 3 if wl == "stress" else (6 if stress else 16)
        ms = [ctx.render(B.default_camera(), w, h, d, want_stats=True)[1].kernel_ms for _ in range(n)]
        ctx.close()
        ctx = make(True)
        st = ctx.render(B.default_camera(), w, h, d, want_stats=True)[1]
        ctx.close()
        rays = max(st.rays, 1)
        print("%-24s %-9s median %9.3f ms  min %9.3f | rays %.3e node tests/ray %.2f tri tests/ray %.2f sphere tests/ray %.2f lanes/node trip %.1f lanes/tri trip %.1f | node trips %.3e tri trips %.3e"
              % (name, wl, statistics.median(ms[min(3, n - 1):]), min(ms[min(3, n - 1):]), rays, st.bvh_node_tests / rays, st.triangle_tests / rays,
                 st.sphere_tests / rays, st.bvh_node_tests / max(st.node_wave_trips, 1), st.triangle_tests / max(st.tri_wave_trips, 1), st.node_wave_trips, st.tri_wave_trips), flush=True)
