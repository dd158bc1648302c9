#!/usr/bin/env python3
"""Benchmark of the trace loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one frame of the reference's default scene (glass, water texture, environment map,
bunny) at 3840x2160, recursion depth 4 -- the configuration BASELINE.json's metric is quoted on.
With N > 1 the same frame is sharded over the ranks by interleaved rows (rank r renders rows
r, r+N, ...), gathered on rank 0 over RCCL and de-interleaved there: total work is fixed, so the
scaling is "strong".  Inputs (scene, textures) are resident in HBM before the timed region; the
output frame stays in HBM.  `--gpus N` without a launcher's WORLD_SIZE in the environment starts the
N ranks itself (torch.distributed.run as a child process, before anything touches a GPU).

value = primary + secondary rays traced per second over the whole job, in Mrays/s, as BASELINE.json
words the metric; a ray is one closest-hit query (one sceneIntersect call from traceRay in the
reference, src/Scene.cpp:87), counted by the kernel itself in a separate statistics pass over the same
frame.  config.mrays_all adds the shadow rays (sceneIntersect calls from calcLightning, :69).
ms_per_step is the steady state of a repeated view (the tile list comes from the previous frame's
costs); first_frame holds what a single frame of a new view costs, moving_camera a view that changes
every frame, host_buffer the rtx_render call that also copies the frame to host memory.

--workload picks another of BASELINE.json's configurations instead (they are parity-test cases in the
first place; the default is the line the metric is quoted on): "planes4k" = spheres + islands + water,
depth 1 (configs[1]), "bunny4k" = the bunny over the water, depth 4 (configs[2]), "8k" = the full scene
at 7680x4320 (configs[3]), "stress" = 10 000 seeded spheres + the bunny subdivided to 1 271 808
triangles, 16384x16384, 4 samples per pixel (configs[4], raytracing_amd/scenes.py).  The CPU baseline is
timed for the 3840x2160 ones.

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

WIDTH, HEIGHT, DEPTH, MASK = 3840, 2160, 4, 31
# name: (width, height, depth, scene mask (1 spheres, 2 boxes, 4 loose triangles, 8 mesh, 16 islands), text, metric suffix)
WORKLOADS = {
    "4k": (3840, 2160, 4, 31, "full default scene, 3840x2160, depth 4, 1 spp (BASELINE.json configs[3] scene at the metric's 4K size)",
           "full scene 4K"),
    "planes4k": (3840, 2160, 1, 17, "spheres + islands + water plane, primary + shadow rays (depth 1), 3840x2160 (BASELINE.json configs[1])",
                 "spheres+planes 4K, depth 1"),
    "bunny4k": (3840, 2160, 4, 8, "bunny.obj mesh over the water plane, depth 4, 3840x2160 (BASELINE.json configs[2])", "bunny 4K"),
    "8k": (7680, 4320, 4, 31, "full default scene, 7680x4320, depth 4, 1 spp (BASELINE.json configs[3])", "full scene 8K"),
    "stress": (16384, 16384, 4, 31, "default scene with 10 000 seeded spheres and the bunny subdivided to 1 271 808 triangles, "
                                    "16384x16384, depth 4, 4 spp (BASELINE.json configs[4])", "stress scene 16K, 4 spp"),
    "stress4k": (4096, 4096, 4, 31, "the stress scene at 4096x4096, depth 4, 4 spp (a size for counter passes)", "stress scene 4K, 4 spp"),
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
            h = lib.ref_scene_create(os.path.join(base, "run").encode(), MASK, os.path.join(base, "resources", "envmap.jpg").encode(),
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
    ctx.load_default_scene(textures["water"], textures["sky"], mask=MASK)
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
            "sample": "one whole frame of the same workload (%dx%d, depth %d) by the reference's own Scene::traceRay " % (WIDTH, HEIGHT, DEPTH) +
                      "(oracle/_ref/libref.so), OpenMP dynamic rows as in Scene::render: %d sceneIntersect calls (it also traces "
                      "the zero-weight children and the shadow rays the GPU prunes) in %.2f s; the bit-equal port took %.2f s"
                      % (rays, ref["seconds"], dt),
            "ms_per_frame": round(ref["seconds"] * 1e3, 1),
            "same_bytes_as_gpu_frame": ref["same_bytes_as_gpu_frame"],
        }
    return {
        "value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": "one whole frame of the same workload (%dx%d, depth %d), OpenMP dynamic rows like the reference: " % (WIDTH, HEIGHT, DEPTH) +
                  "%d sceneIntersect calls (the reference also traces the zero-weight children the GPU prunes) in %.2f s"
                  % (rays, dt),
        "ms_per_frame": round(dt * step * 1e3, 1),
    }


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(n):
    """`--gpus n` without a launcher: start the n ranks as a child (torch.distributed.run), before this
    process has touched a GPU, and leave with the child's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def moved(cam, k):
    """The default camera after k small steps along x (a view that changes every frame)."""
    import copy
    c = copy.copy(cam)
    c.origin[0] = cam.origin[0] + 0.002 * k
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="4k")
    args = ap.parse_args()
    global WIDTH, HEIGHT, DEPTH, MASK
    WIDTH, HEIGHT, DEPTH, MASK, workload_text, metric_text = WORKLOADS[args.workload]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)

    import numpy as np
    import torch
    from raytracing_amd.distributed import ShardedRenderer
    from raytracing_amd.rtx import ASSETS_DIR, OPT_SCHEDULE_FEEDBACK, OPT_STATS, Frame, hip_backend

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
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
    stress = args.workload.startswith("stress")

    def make_ctx(stats=False, options=()):
        ctx = hip.create(local_rank)
        if stress:
            from raytracing_amd import scenes
            scenes.load_stress_scene(ctx, hip, textures["water"], textures["sky"])
        else:
            ctx.load_default_scene(textures["water"], textures["sky"], mask=MASK)
        if stats:
            ctx.set_option(OPT_STATS, 1)
        for opt, val in options:
            ctx.set_option(opt, val)
        ctx.commit()
        return ctx

    ctx = make_ctx()
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
    sctx = make_ctx(stats=True)
    fr = Frame(WIDTH, HEIGHT, DEPTH, rank, world)
    rows = sctx.frame_rows(fr)
    buf = torch.zeros(rows * WIDTH * 3, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    sctx.render_device(cam, fr, buf.data_ptr(), renderer.stream.cuda_stream)
    torch.cuda.synchronize()
    st = sctx.stats()
    sctx.close()
    counts = torch.tensor([st.primary_rays, st.secondary_rays, st.shadow_rays], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(counts)
    primary, secondary, shadow = (int(x) for x in counts.tolist())
    rays_per_frame = primary + secondary + shadow

    # ---- dominant kernel: live HIP-event timing of k_render on its own stream (rank 0's share) ----
    roofline = None
    extra = {}
    if rank == 0:
        hipapi = hip_event_api()
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        hipapi.hipEventCreate(ctypes.byref(e0))
        hipapi.hipEventCreate(ctypes.byref(e1))
        stream = renderer.stream.cuda_stream  # the stream the kernel is launched on

        def timed(fn, n):
            """ms per call of fn over n calls, HIP events on `stream`"""
            hipapi.hipEventRecord(e0, ctypes.c_void_p(stream))
            for k in range(n):
                fn(k)
            hipapi.hipEventRecord(e1, ctypes.c_void_p(stream))
            hipapi.hipEventSynchronize(e1)
            ms = ctypes.c_float()
            hipapi.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
            return ms.value / n

        heavy = stress  # a large fraction of a second per frame, or seconds
        n_launch = max(2, min(args.steps, 5)) if heavy else max(20, min(args.steps, 200))
        for _ in range(1 if heavy else 5):
            ctx.render_device(cam, fr, buf.data_ptr(), stream)
        kernel_ms = timed(lambda k: ctx.render_device(cam, fr, buf.data_ptr(), stream), n_launch)
        flops = algorithmic_flops(st, rows * WIDTH)
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        # counters of the same kernel on the same frame, from the committed rocprofv3 passes
        traffic, issue = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                prof = json.load(open(tpath)).get(args.workload, {})
                traffic = prof.get("k_render_hbm_bytes_per_launch_n%d" % world)
                wave_insts = prof.get("k_render_valu_wave_insts_per_launch_n%d" % world)
                lane_insts = prof.get("k_render_valu_lane_insts_per_launch_n%d" % world)
                if wave_insts and lane_insts:
                    # one wave64 VALU instruction occupies its SIMD for 4 clocks; 1024 SIMDs; 64 lanes/CU/clk
                    issue = {"valu_wave_insts_per_launch": int(wave_insts), "valu_lane_insts_per_launch": int(lane_insts),
                             "simd_issue_frac": round(wave_insts * 4 / (N_SIMD * CLOCK_HZ * kernel_ms * 1e-3), 4),
                             "lane_slot_frac": round(lane_insts / (N_SIMD * 16 * CLOCK_HZ * kernel_ms * 1e-3), 4),
                             "mean_active_lanes": round(lane_insts / wave_insts, 1),
                             "source": "%s (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU) / live kernel time" % prof.get("source", "profiles/traffic.json")}
            except Exception:
                traffic, issue = None, None
        # algorithmic bytes: BVH nodes 64 B, triangles 48 B, texels 3 B, framebuffer 3 B/pixel
        alg_bytes = st.bvh_node_tests * 64 + st.triangle_tests * 48 + (st.sky_lookups + st.plane_tests) * 3 + rows * WIDTH * 3
        roofline = {
            "bound": "valu", "kernel": "k_wf_trace + k_wf_shadow + k_wf_shade<.., true> (level-by-level pipeline, all launches of a frame)" if stress else "k_render<false, false>", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 5), "traffic": traffic,
            "kernel_ms": round(kernel_ms, 4), "algorithmic_flops_per_launch": int(flops),
            "note": "FP32 vector (not MFMA: no dense contraction). Pixel arithmetic is unfused and unpacked to track "
                    "the reference bit for bit, so 1 flop/lane/clk = 39.3 TFLOP/s is the most such code can reach",
            "valu_issue": issue,
            "hbm": {"algorithmic_bytes_per_launch": int(alg_bytes), "achieved_gbs": round(alg_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                    "peak_gbs": PEAK_HBM_GBS, "frac": round(alg_bytes / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 5)},
        }

        if not heavy:
            # ---- what the steady state leaves out (this rank's share; device time by HIP events) ----
            # a single frame of a new view: fresh contexts, the very first frame each (tile list from projected boxes)
            firsts = []
            for _ in range(5):
                c1 = make_ctx()
                torch.cuda.synchronize()
                firsts.append(timed(lambda k: c1.render_device(cam, fr, buf.data_ptr(), stream), 1))
                c1.close()
            # the same view again and again without the feedback (every frame scheduled like a first frame)
            c2 = make_ctx(options=((OPT_SCHEDULE_FEEDBACK, 0),))
            for _ in range(3):
                c2.render_device(cam, fr, buf.data_ptr(), stream)
            static_ms = timed(lambda k: c2.render_device(cam, fr, buf.data_ptr(), stream), 20)
            c2.close()
            # a camera that moves every frame (the list is rebuilt from the previous frame's costs)
            for k in range(5):
                ctx.render_device(moved(cam, k), fr, buf.data_ptr(), stream)
            moving_ms = timed(lambda k: ctx.render_device(moved(cam, 5 + k), fr, buf.data_ptr(), stream), 50)
            # rtx_render: the same frame into a host buffer (adds the device-to-host copy and a stream synchronisation)
            for _ in range(3):
                ctx.render(cam, WIDTH, HEIGHT, DEPTH, rank, world)
            t1 = time.perf_counter()
            for _ in range(10):
                ctx.render(cam, WIDTH, HEIGHT, DEPTH, rank, world)
            host_ms = (time.perf_counter() - t1) / 10 * 1e3
            extra = {"first_frame": {"ms": round(float(np.median(firsts)), 4), "min_ms": round(min(firsts), 4), "max_ms": round(max(firsts), 4),
                                     "what": "device time of the first frame on a fresh context (5 contexts): tile classes from projected "
                                             "boxes + list build + k_render"},
                     "static_order_ms": round(static_ms, 4),
                     "moving_camera_ms": round(moving_ms, 4),
                     "host_buffer": {"ms": round(host_ms, 4), "what": "wall clock of rtx_render (frame copied to pageable host memory), repeated view"}}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s (primary+secondary) & ms/frame, " + metric_text,
            "value": round((primary + secondary) / (ms_per_step * 1e-3) / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "the reference's own scene constants and assets (bunny.obj, water.jpg, envmap.jpg)"
                    + ("; spheres from PCG32 seed 0x5EED" if stress else "; no randomness"),
            "config": {"workload": workload_text, "width": WIDTH, "height": HEIGHT, "depth": DEPTH,
                       "parallelism": "rows interleaved over %d GPU(s)%s" % (world, ", RCCL gather to rank 0" if world > 1 else ""),
                       "frames": "ms_per_step is the steady state of one view rendered again and again: from the second frame on the "
                                 "tile list comes from the previous frame's measured tile costs; first_frame / moving_camera_ms / "
                                 "host_buffer give the other cases",
                       "rays_per_frame": rays_per_frame, "primary": primary, "secondary": secondary, "shadow": shadow,
                       "mrays_all": round(rays_per_frame / (ms_per_step * 1e-3) / 1e6, 2)},
            "roofline": roofline,
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline and (WIDTH, HEIGHT) == (3840, 2160) and not stress:
            gpu_frame = renderer.render(cam)[0]
            renderer.finish()
            torch.cuda.synchronize()
            out["cpu_baseline"] = cpu_baseline(textures, lambda b: b.default_camera(),
                                               gpu_frame.cpu().numpy().reshape(HEIGHT, WIDTH, 3))
            if "first_frame" in out:
                out["cpu_baseline"]["gpu_first_frame_speedup"] = round(out["cpu_baseline"]["ms_per_frame"] / out["first_frame"]["ms"], 1)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
