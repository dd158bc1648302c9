#!/usr/bin/env python3
"""Benchmark of the trace loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one frame of the reference's default scene (glass, water texture, environment map,
bunny) at 3840x2160, recursion depth 4 -- the configuration BASELINE.json's metric is quoted on.
With N > 1 the same frame is sharded over the ranks by interleaved rows (rank r renders rows
r, r+N, ...), gathered on rank 0 over RCCL and de-interleaved there: total work is fixed, so the
scaling is "strong".  Inputs (scene, textures) are resident in HBM before the timed region; the
output frame stays in HBM.

value = rays traced per second over the whole job, in Mrays/s; a ray is one closest-hit or shadow
query (one sceneIntersect call in the reference, src/Scene.cpp:29), counted by the kernel itself in
a separate statistics pass over the same frame.

--workload picks another of BASELINE.json's configurations instead (they are parity-test cases in the
first place; the default is the line the metric is quoted on): "8k" = the same scene at 7680x4320,
"stress" = 10 000 seeded spheres + the bunny subdivided to 1 271 808 triangles, 16384x16384, 4 samples
per pixel (raytracing_amd/scenes.py).  No CPU baseline is timed for those.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, DEPTH = 3840, 2160, 4
WORKLOADS = {
    "4k": (3840, 2160, "full default scene, 3840x2160, depth 4, 1 spp (BASELINE.json configs[3] scene at the metric's 4K size)"),
    "8k": (7680, 4320, "full default scene, 7680x4320, depth 4, 1 spp (BASELINE.json configs[3])"),
    "stress": (16384, 16384, "default scene with 10 000 seeded spheres and the bunny subdivided to 1 271 808 triangles, "
                             "16384x16384, depth 4, 4 spp (BASELINE.json configs[4])"),
    "stress4k": (4096, 4096, "the stress scene at 4096x4096, depth 4, 4 spp (a size for counter passes)"),
}

# Algorithmic FP32 operation counts per unit of work (mul, add, sub, compare-free; div, sqrt, rcp
# count 1 each), per unit of work as the kernels do it.
FLOPS = {
    "primary_ray": 28,        # ray generation + normalize
    "sphere_test": 19,
    "tri_test_early": 26,     # Moller-Trumbore up to the first (u) rejection
    "tri_test_rest": 24,      # ... the remainder of a full test (50 in total)
    "box_test": 18,           # slab test of one box
    "bvh_node_test": 36,      # two child boxes
    "plane_test": 5,          # ray/plane
    "island_test": 18,        # ellipse containment (plane + ellipse = 23)
    "water_shade": 23,        # texel blend of a water hit (plane + blend = 28)
    "light_setup": 26,        # per shadow ray: normalize, n.l, offset origin, distance
    "light_phong": 29,        # per unoccluded light: reflect, dots, powf (counted as 1 op ~ 20)
    "sky_lookup": 30,
    "pixel": 6,               # clamp + scale of three channels
}
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X FP32 vector peak (packed FMA); = the f32 MFMA peak
PEAK_HBM_GBS = 8000.0
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMD16; 157.3 TFLOP/s = 1024 x 16 lanes x 2 (FMA) x 2 (packed) x 2.4 GHz


def algorithmic_flops(st, n_pixels):
    f = FLOPS
    return (st.primary_rays * f["primary_ray"] + st.sphere_tests * f["sphere_test"]
            + (st.triangle_tests + st.loose_tri_tests) * f["tri_test_early"] + st.triangle_full * f["tri_test_rest"]
            + st.box_tests * f["box_test"] + st.bvh_node_tests * f["bvh_node_test"] + st.plane_tests * f["plane_test"]
            + st.island_tests * f["island_test"] + st.primary_rays * 0 + st.shadow_rays * f["light_setup"]
            + st.lit_lights * f["light_phong"] + st.sky_lookups * f["sky_lookup"] + n_pixels * f["pixel"])


def hip_event_api():
    """HIP events through libamdhip64 directly, recorded on the stream the kernel is launched on."""
    lib = ctypes.CDLL("libamdhip64.so")
    lib.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    lib.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.hipEventSynchronize.argtypes = [ctypes.c_void_p]
    lib.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    return lib


def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    return min(cores, omp) if omp else cores


def reference_baseline(gpu_frame):
    """The reference's own code (oracle/_ref/libref.so: its unmodified sources compiled where they lie,
    plus our extern "C" harness around Scene::traceRay) on one whole frame of the same workload.
    Returns None when that library has not been built or cannot run here."""
    import tempfile
    import numpy as np
    import oracle
    if not os.path.exists(oracle.REF_LIB_PATH):
        return None
    try:
        lib = ctypes.CDLL(oracle.REF_LIB_PATH)
        lib.ref_scene_create.restype = ctypes.c_void_p
        lib.ref_scene_create.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_char_p, ctypes.c_char_p]
        lib.ref_render.restype = ctypes.c_double
        lib.ref_render.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        # the reference opens "../resources/bunny.obj" relative to its working directory
        # (and "../resources/map2.jpg", its default environment map, a copy of envmap.jpg)
        base = tempfile.mkdtemp(prefix="rtx_ref_")
        os.mkdir(os.path.join(base, "run"))
        os.mkdir(os.path.join(base, "resources"))
        for name, src in (("bunny.obj", "bunny.obj"), ("water.jpg", "water.jpg"), ("envmap.jpg", "envmap.jpg"),
                          ("map2.jpg", "envmap.jpg"), ("map.jpg", "map.jpg")):
            os.symlink(os.path.join(ROOT, "assets", src), os.path.join(base, "resources", name))
        # the reference prints to stdout; this program's stdout carries exactly one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            h = lib.ref_scene_create(os.path.join(base, "run").encode(), 31, os.path.join(base, "resources", "envmap.jpg").encode(),
                                     os.path.join(base, "resources", "water.jpg").encode())
            if not h:
                return None
            cores = host_cores()
            frame = np.zeros((HEIGHT, WIDTH, 3), np.uint8)
            dt = lib.ref_render(ctypes.c_void_p(h), WIDTH, HEIGHT, DEPTH, cores, frame.ctypes.data)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    except Exception as e:  # missing library dependencies, a different host ...
        print("reference baseline unavailable: %r" % (e,), file=sys.stderr)
        return None
    return {"seconds": dt, "cores": cores, "same_bytes_as_gpu_frame": bool(gpu_frame is not None and np.array_equal(frame, gpu_frame))}


def cpu_baseline(textures, cam_of, gpu_frame=None):
    """One whole frame of the same workload on the host cores: the reference's own code when
    oracle/_ref/libref.so is there (kind "reference"), else the oracle (CPU restatement of the reference,
    bit-equal to it; kind "port").  Ray counts come from the oracle's statistics either way."""
    import oracle
    from raytracing_amd.rtx import OPT_STATS, Backend
    cpu = Backend(oracle.ORACLE_LIB_PATH)
    ctx = cpu.create()
    ctx.load_default_scene(textures["water"], textures["sky"])
    ctx.set_option(OPT_STATS, 1)
    ctx.commit()
    step = 1
    t0 = time.time()
    _, st = ctx.render(cam_of(cpu), WIDTH, HEIGHT, DEPTH, row_first=0, row_step=step, want_stats=True)
    dt = time.time() - t0
    rays = st.rays
    cores = host_cores()
    ref = reference_baseline(gpu_frame)
    if ref is not None:
        return {
            "value": round(rays / ref["seconds"] / 1e6, 3), "unit": "Mrays/s", "cores": ref["cores"], "kind": "reference",
            "sample": "one whole frame of the same workload (3840x2160, depth 4) by the reference's own Scene::traceRay "
                      "(oracle/_ref/libref.so), OpenMP dynamic rows as in Scene::render: %d sceneIntersect calls (it also traces "
                      "the zero-weight children and the shadow rays the GPU prunes) in %.2f s; the bit-equal port took %.2f s"
                      % (rays, ref["seconds"], dt),
            "ms_per_frame": round(ref["seconds"] * 1e3, 1),
            "same_bytes_as_gpu_frame": ref["same_bytes_as_gpu_frame"],
        }
    return {
        "value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "one whole frame of the same workload (3840x2160, depth 4), OpenMP dynamic rows like the reference: "
                  "%d sceneIntersect calls (the reference also traces the zero-weight children the GPU prunes) in %.2f s"
                  % (rays, dt),
        "ms_per_frame": round(dt * step * 1e3, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="4k")
    args = ap.parse_args()
    global WIDTH, HEIGHT
    WIDTH, HEIGHT, workload_text = WORKLOADS[args.workload]

    import torch
    from raytracing_amd.distributed import ShardedRenderer
    from raytracing_amd.rtx import ASSETS_DIR, OPT_STATS, Frame, hip_backend

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    hip = hip_backend()  # raises if librtx_hip.so is missing: there is no other path
    textures = {"water": hip.decode_jpeg(os.path.join(ASSETS_DIR, "water.jpg")),
                "sky": hip.decode_jpeg(os.path.join(ASSETS_DIR, "envmap.jpg"))}
    cam = hip.default_camera()

    def make_ctx(stats):
        ctx = hip.create(local_rank)
        if args.workload.startswith("stress"):
            from raytracing_amd import scenes
            scenes.load_stress_scene(ctx, hip, textures["water"], textures["sky"])
        else:
            ctx.load_default_scene(textures["water"], textures["sky"])
        if stats:
            ctx.set_option(OPT_STATS, 1)
        ctx.commit()
        return ctx

    ctx = make_ctx(False)
    renderer = ShardedRenderer(ctx, WIDTH, HEIGHT, DEPTH, rank, world, torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        renderer.render(cam)
    renderer.finish()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        renderer.render(cam)
    renderer.finish()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- rays per frame: statistics pass over this rank's rows, summed over ranks ----
    sctx = make_ctx(True)
    fr = Frame(WIDTH, HEIGHT, DEPTH, rank, world)
    rows = sctx.frame_rows(fr)
    buf = torch.zeros(rows * WIDTH * 3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    sctx.render_device(cam, fr, buf.data_ptr(), renderer.stream.cuda_stream)
    torch.cuda.synchronize()
    st = sctx.stats()
    counts = torch.tensor([st.primary_rays, st.secondary_rays, st.shadow_rays], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(counts)
    primary, secondary, shadow = (int(x) for x in counts.tolist())
    rays_per_frame = primary + secondary + shadow

    # ---- dominant kernel: live HIP-event timing of k_render on its own stream (rank 0's share) ----
    roofline = None
    if rank == 0:
        hipapi = hip_event_api()
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        hipapi.hipEventCreate(ctypes.byref(e0))
        hipapi.hipEventCreate(ctypes.byref(e1))
        stream = renderer.stream.cuda_stream  # the stream the kernel is launched on
        heavy = args.workload.startswith("stress")  # a large fraction of a second per frame, or seconds
        n_launch = max(2, min(args.steps, 5)) if heavy else max(20, min(args.steps, 200))
        for _ in range(1 if heavy else 5):
            ctx.render_device(cam, fr, buf.data_ptr(), stream)
        hipapi.hipEventRecord(e0, ctypes.c_void_p(stream))
        for _ in range(n_launch):
            ctx.render_device(cam, fr, buf.data_ptr(), stream)
        hipapi.hipEventRecord(e1, ctypes.c_void_p(stream))
        hipapi.hipEventSynchronize(e1)
        ms = ctypes.c_float()
        hipapi.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
        kernel_ms = ms.value / n_launch
        flops = algorithmic_flops(st, rows * WIDTH)
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        # counters of the same kernel on the same frame, from the committed rocprofv3 passes
        traffic, issue = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                prof = json.load(open(tpath))
                # (the committed counter passes are of the default workload)
                traffic = prof.get("k_render_hbm_bytes_per_launch_n%d" % world) if args.workload == "4k" else None
                wave_insts = prof.get("k_render_valu_wave_insts_per_launch_n%d" % world) if args.workload == "4k" else None
                lane_insts = prof.get("k_render_valu_lane_insts_per_launch_n%d" % world) if args.workload == "4k" else None
                if wave_insts and lane_insts:
                    # one wave64 VALU instruction occupies its SIMD for 4 clocks; 1024 SIMDs; 64 lanes/CU/clk
                    issue = {"valu_wave_insts_per_launch": int(wave_insts), "valu_lane_insts_per_launch": int(lane_insts),
                             "simd_issue_frac": round(wave_insts * 4 / (N_SIMD * CLOCK_HZ * kernel_ms * 1e-3), 4),
                             "lane_slot_frac": round(lane_insts / (N_SIMD * 16 * CLOCK_HZ * kernel_ms * 1e-3), 4),
                             "mean_active_lanes": round(lane_insts / wave_insts, 1),
                             "source": "profiles/traffic.json (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU) / live kernel time"}
            except Exception:
                traffic, issue = None, None
        # algorithmic bytes: BVH nodes 64 B, triangles 48 B, texels 3 B, framebuffer 3 B/pixel
        alg_bytes = st.bvh_node_tests * 64 + st.triangle_tests * 48 + (st.sky_lookups + st.plane_tests) * 3 + rows * WIDTH * 3
        roofline = {
            "bound": "valu", "kernel": "k_render<false, true>" if args.workload.startswith("stress") else "k_render<false, false>", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 5), "traffic": traffic,
            "kernel_ms": round(kernel_ms, 4), "algorithmic_flops_per_launch": int(flops),
            "note": "FP32 vector (not MFMA: no dense contraction). Pixel arithmetic is unfused and unpacked to track "
                    "the reference bit for bit, so 1 flop/lane/clk = 39.3 TFLOP/s is the most such code can reach",
            "valu_issue": issue,
            "hbm": {"algorithmic_bytes_per_launch": int(alg_bytes), "achieved_gbs": round(alg_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                    "peak_gbs": PEAK_HBM_GBS, "frac": round(alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)},
        }

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s (primary+secondary+shadow rays traced) & ms/frame, " + {"4k": "full scene 4K", "8k": "full scene 8K", "stress": "stress scene 16K, 4 spp",
                                                                                         "stress4k": "stress scene 4K, 4 spp"}[args.workload],
            "value": round(rays_per_frame / (ms_per_step * 1e-3) / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "the reference's own scene constants and assets (bunny.obj, water.jpg, envmap.jpg)"
                    + ("; spheres from PCG32 seed 0x5EED" if args.workload.startswith("stress") else "; no randomness"),
            "config": {"workload": workload_text, "width": WIDTH, "height": HEIGHT, "depth": DEPTH,
                       "parallelism": "rows interleaved over %d GPU(s)%s" % (world, ", RCCL gather to rank 0" if world > 1 else ""),
                       "rays_per_frame": rays_per_frame, "primary": primary, "secondary": secondary, "shadow": shadow,
                       "mrays_primary_secondary": round((primary + secondary) / (ms_per_step * 1e-3) / 1e6, 2)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "4k":
            gpu_frame = renderer.render(cam)[0]
            renderer.finish()
            torch.cuda.synchronize()
            out["cpu_baseline"] = cpu_baseline(textures, lambda b: b.default_camera(),
                                               gpu_frame.cpu().numpy().reshape(HEIGHT, WIDTH, 3))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
