"""Renders FRAMES frames of one bench workload through the C ABI and nothing else (no torch: starts in a second).
bench.py runs it under `rocprofv3 --pmc ...`, one counter group per run, and reads the counters of the render
kernels from the CSV.  usage: python3 scripts/counter_frames.py <workload> [frames] [library]
(RTX_COUNTER_OPTS="8=0,7=0" in the environment sets options of the context: id=value pairs of rtx_set_option;
for the stress workloads RTX_STRESS_LEVELS=n subdivides the mesh n times instead of four and RTX_STRESS_SKY=k keeps every
k-th row and column of the panorama, RTX_STRESS_WATER=k of the water texture -- to see what part of the traffic is the mesh
and what part the textures)"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402  (the workload table)
from raytracing_amd import scenes  # noqa: E402
from raytracing_amd.rtx import ASSETS_DIR, Backend, HIP_LIB_PATH  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "4k"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H = Backend(sys.argv[3] if len(sys.argv) > 3 else HIP_LIB_PATH)
w, h, depth, mask = bench.WORKLOADS[name][:4]
water, sky = H.decode_jpeg(ASSETS_DIR + "/water.jpg"), H.decode_jpeg(ASSETS_DIR + "/envmap.jpg")
ctx = H.create()
if name.startswith("stress"):
    k = int(os.environ.get("RTX_STRESS_SKY", "1"))
    if k > 1:
        import numpy as np
        sky = np.ascontiguousarray(sky[::k, ::k])
    k = int(os.environ.get("RTX_STRESS_WATER", "1"))
    if k > 1:
        import numpy as np
        water = np.ascontiguousarray(water[::k, ::k])
    scenes.load_stress_scene(ctx, H, water, sky, levels=int(os.environ.get("RTX_STRESS_LEVELS", "4")))
else:
    ctx.load_default_scene(water, sky, mask=mask)
for pair in filter(None, os.environ.get("RTX_COUNTER_OPTS", "").split(",")):
    ctx.set_option(int(pair.split("=")[0]), float(pair.split("=")[1]))
ctx.commit()
cam = H.default_camera()
for _ in range(frames):
    ctx.render(cam, w, h, depth)
ctx.close()
print("frames", frames, flush=True)
