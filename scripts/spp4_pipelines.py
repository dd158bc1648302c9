"""The two pipelines at four samples per pixel on the reference's scenes (full scene, mesh only; run on a GPU box):
python scripts/spp4_pipelines.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd.rtx import hip_backend, ASSETS_DIR, OPT_SPP, OPT_PIPELINE
H = hip_backend()
water, sky = H.decode_jpeg(ASSETS_DIR + '/water.jpg'), H.decode_jpeg(ASSETS_DIR + '/envmap.jpg')
cam = H.default_camera()
for (w, h) in ((3840, 2160), (1920, 1080)):
    for mask in (31, 8):
        for pipe in (0, 1):
            ctx = H.create(); ctx.load_default_scene(water, sky, mask=mask); ctx.set_option(OPT_SPP, 4); ctx.set_option(OPT_PIPELINE, pipe); ctx.commit()
            ms = sorted(ctx.render(cam, w, h, 4, want_stats=True)[1].kernel_ms for _ in range(8))
            print("%dx%d mask %d spp 4 pipeline %d: %.3f ms" % (w, h, mask, pipe, ms[1]), flush=True)
            ctx.close()
