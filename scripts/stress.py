"""Kernel time and ray counts of the stress scene (10 000 seeded spheres, the bunny subdivided to
1 271 808 triangles, 4 samples per pixel) at 4096^2, 8192^2 and 16384^2 (run on a GPU box).
Prints one JSON line per size."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from raytracing_amd import scenes
from raytracing_amd.rtx import ASSETS_DIR, OPT_STATS, hip_backend

H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
cam = H.default_camera()
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]
for stats in (0, 1):
    ctx = H.create()
    t0 = time.time()
    nt = scenes.load_stress_scene(ctx, H, water, sky)
    ctx.set_option(OPT_STATS, stats)
    t1 = time.time()
    ctx.commit()
    t2 = time.time()
    for size in sizes:
        ms = []
        for _ in range(1 if stats else 4):
            _, st = ctx.render(cam, size, size, 4, want_stats=True)
            ms.append(st.kernel_ms)
        if stats:
            print(json.dumps({"size": size, "rays": st.rays, "primary": st.primary_rays, "secondary": st.secondary_rays,
                              "shadow": st.shadow_rays, "sphere_tests": st.sphere_tests, "bvh_node_tests": st.bvh_node_tests,
                              "triangle_tests": st.triangle_tests, "lanes_per_node_trip": round(st.bvh_node_tests / max(1, st.node_wave_trips), 1), "stats_ms": round(ms[0], 2)}), flush=True)
        else:
            print(json.dumps({"size": size, "triangles": nt, "spheres": scenes.STRESS_SPHERES, "spp": 4, "build_arrays_s": round(t1 - t0, 2),
                              "commit_s": round(t2 - t1, 2), "first_frame_ms": round(ms[0], 2), "later_frames_ms": [round(m, 2) for m in ms[1:]]}),
                  flush=True)
