"""What the tiles of a frame cost: builds a copy of the library that writes the per-tile cost array (100 MHz ticks per
8x8 tile, what the schedule feedback goes by) of every frame to $RTX_TILE_COST_FILE, renders a workload twice and prints
the distribution of the second frame's costs and where the dearest tiles lie.  The probe lives in a temporary copy of
the sources, never in the product.
usage: python scripts/tile_costs.py build   (here, needs hipcc)
       python scripts/tile_costs.py run [4k]   (on a GPU box, after build)"""
import os
import shutil
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(R, "raytracing_amd", "librtx_hip_tile_costs.so")  # git-ignored like every built library


def build():
    tmp = tempfile.mkdtemp(prefix="rtx_tile_costs_")
    os.makedirs(os.path.join(tmp, "raytracing_amd"))
    csrc = os.path.join(tmp, "raytracing_amd", "csrc")
    shutil.copytree(os.path.join(R, "raytracing_amd", "csrc"), csrc)
    os.symlink(os.path.join(R, "include"), os.path.join(tmp, "include"))
    p = os.path.join(csrc, "rtx_hip.hip")
    s = open(p).read()
    old = "    ss.heads_flip = !ss.heads_flip;  // only now: a launch that failed has cleared nothing"
    if old not in s:
        raise SystemExit("tile_costs: the sources have changed")
    s = s.replace(old, """    if (const char *path = getenv("RTX_TILE_COST_FILE")) {
        std::vector<unsigned> host((size_t)n_tiles);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(host.data(), m.cost, host.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
        if (FILE *fp = fopen(path, "wb")) {
            const unsigned hdr[2] = {(unsigned)tiles_x, (unsigned)tiles_y};
            fwrite(hdr, sizeof hdr, 1, fp), fwrite(host.data(), sizeof(unsigned), host.size(), fp), fclose(fp);
        }
    }
""" + old, 1)
    open(p, "w").write(s)
    host = [os.path.join(csrc, "host", f) for f in sorted(os.listdir(os.path.join(csrc, "host"))) if f.endswith(".cpp") and f != "rt_main.cpp"]
    cmd = [shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
           "-fno-slp-vectorize", "-I" + os.path.join(R, "include"), "-ffp-contract=off", "-shared", "-Wl,-Bsymbolic",
           os.path.join(csrc, "rtx_hip.hip"), os.path.join(csrc, "rtx_group.cpp")] + host + ["-ldl", "-o", LIB]
    subprocess.check_call(cmd)
    shutil.rmtree(tmp, ignore_errors=True)
    print("built", LIB)


def run(workloads):
    import numpy as np
    out = os.path.join(R, "gpurun_out", "tile_cost.bin")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    os.environ["RTX_TILE_COST_FILE"] = out
    for w in workloads:
        subprocess.check_call([sys.executable, os.path.join(R, "scripts", "counter_frames.py"), w, "2", LIB])
        raw = np.fromfile(out, dtype=np.uint32)
        tx, ty = int(raw[0]), int(raw[1])
        c = raw[2:2 + tx * ty].astype(np.float64) / 100.0  # us
        print("== %s: %d x %d tiles, sum %.0f us (%.1f us per tile), max %.0f us" % (w, tx, ty, c.sum(), c.mean(), c.max()))
        srt = np.sort(c)[::-1]
        for k in (1, 10, 100, 1000, 3000, 10000, 30000, 100000):
            if k <= len(srt):
                print("   rank %6d: %7.1f us; the %d dearest tiles hold %.1f %% of all cost" % (k, srt[k - 1], k, 100 * srt[:k].sum() / c.sum()))
        grid = c.reshape(ty, tx)
        top = np.argsort(c)[::-1][:3000]
        ys, xs = top // tx, top % tx
        print("   the 3000 dearest tiles lie in tile rows %d..%d, tile columns %d..%d" % (ys.min(), ys.max(), xs.min(), xs.max()))
        # a coarse map: mean cost per 16 x 16 block of tiles
        by, bx = (ty + 15) // 16, (tx + 15) // 16
        print("   mean cost (us) per block of 16 x 16 tiles:")
        for j in range(by):
            print("   " + " ".join("%5.0f" % grid[16 * j:16 * j + 16, 16 * i:16 * i + 16].mean() for i in range(bx)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:] or ["4k"])
