"""Where a wave of k_render spends its time: builds a copy of the library with wave-clock probes around the
regions of the path loop (taking a tile, the filters and small lists before the mesh, the mesh walk, planes /
islands / hit record, shading and path bookkeeping) and prints the shares per frame, plus how many lanes have a
ray to cast on an average trip of the loop.  The probes live in a temporary copy of the sources, never in the
product.  usage: python scripts/region_clocks.py build   (here, needs hipcc)
                 python scripts/region_clocks.py run [4k planes4k bunny4k]   (on a GPU box, after build)"""
import os
import shutil
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(R, "raytracing_amd", "librtx_hip_regions.so")  # git-ignored like every built library


def patch(text, old, new, count=1):
    if text.count(old) < 1:
        raise SystemExit("region_clocks: the sources have changed, this piece is gone:\n" + old)
    return text.replace(old, new, count)


def build():
    tmp = tempfile.mkdtemp(prefix="rtx_regions_")
    os.makedirs(os.path.join(tmp, "raytracing_amd"))
    csrc = os.path.join(tmp, "raytracing_amd", "csrc")
    shutil.copytree(os.path.join(R, "raytracing_amd", "csrc"), csrc)
    os.symlink(os.path.join(R, "include"), os.path.join(tmp, "include"))
    p = os.path.join(csrc, "rtx_trace.hpp")
    s = open(p).read()
    s = patch(s, "template <>\nstruct LaneCounters<false> {};\n", """template <>
struct LaneCounters<false> {
    long long last = 0;
    long long acc[6] = {0, 0, 0, 0, 0, 0};
    long long iters = 0, lanes = 0;
};
template <int K>
RTX_DEV void regionEnd(LaneCounters<false> &lc) {
    const long long now = __builtin_readcyclecounter();
    lc.acc[K] += now - lc.last;
    lc.last = now;
}
template <int K>
RTX_DEV void regionEnd(LaneCounters<true> &) {}
""")
    s = patch(s, """    CastState cs;
    if (castBegin<STATS, WIDE>(s, q, orig, dir, t_exit, cs, stack, stride, lc)) castMesh<STATS>(s, orig, dir, t_exit, cs, stack, stride, lc);
    return castEnd<STATS, WIDE>(s, q, orig, dir, shadow, t_exit, cs, h, min_dist, lc);""", """    CastState cs;
    regionEnd<0>(lc);
    const bool need = castBegin<STATS, WIDE>(s, q, orig, dir, t_exit, cs, stack, stride, lc);
    regionEnd<1>(lc);
    if (need) castMesh<STATS>(s, orig, dir, t_exit, cs, stack, stride, lc);
    regionEnd<2>(lc);
    const bool r = castEnd<STATS, WIDE>(s, q, orig, dir, shadow, t_exit, cs, h, min_dist, lc);
    regionEnd<3>(lc);
    return r;""")
    open(p, "w").write(s)
    p = os.path.join(csrc, "rtx_hip.hip")
    s = open(p).read()
    s = patch(s, """template <bool STATS>
RTX_DEV void flushCounters(const LaneCounters<STATS> &lc, Counters *out) {
    if constexpr (STATS) {""", """__device__ unsigned long long g_region[6];
__device__ unsigned long long g_iters[2];
template <bool STATS>
RTX_DEV void flushCounters(const LaneCounters<STATS> &lc, Counters *out) {
    if constexpr (!STATS) {
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 6; ++k) atomicAdd(&g_region[k], (unsigned long long)lc.acc[k]);
        if (lc.iters) atomicAdd(&g_iters[0], (unsigned long long)lc.iters), atomicAdd(&g_iters[1], (unsigned long long)lc.lanes);
    }
    if constexpr (STATS) {""")
    s = patch(s, """        float md = 0.0f;
        if (pathCasts(st, quad_j)) {
            pathCountRay<STATS>(st, quad_j, lc);""", """        float md = 0.0f;
        if constexpr (!STATS) {
            const unsigned long long casting = __ballot(pathCasts(st, quad_j));
            if (firstActiveLane()) lc.iters += 1, lc.lanes += __popcll(casting);
        }
        if (pathCasts(st, quad_j)) {
            pathCountRay<STATS>(st, quad_j, lc);""")
    s = patch(s, """        V3 c;
        if (pathConsume<STATS, WIDE>(s, q, st, frames, hit, h, md, quad_j, c, lc)) {
            if constexpr (WIDE) {""", """        V3 c;
        const bool path_done = pathConsume<STATS, WIDE>(s, q, st, frames, hit, h, md, quad_j, c, lc);
        regionEnd<4>(lc);
        if (path_done) {
            if constexpr (WIDE) {""")
    s = patch(s, """    LaneCounters<STATS> lc;
    PathState st;
    FrameSpill spill;""", """    LaneCounters<STATS> lc;
    if constexpr (!STATS) lc.last = __builtin_readcyclecounter();
    PathState st;
    FrameSpill spill;""")
    s = patch(s, """        // ---- take a tile ----
        unsigned tile = 0xffffffffu;""", """        // ---- take a tile ----
        regionEnd<0>(lc);
        unsigned tile = 0xffffffffu;""")
    s = patch(s, """        if (tile == 0xffffffffu) break;  // every queue is empty""", """        regionEnd<5>(lc);
        if (tile == 0xffffffffu) break;  // every queue is empty""")
    s = patch(s, """    ss.heads_flip = !ss.heads_flip;  // only now: a launch that failed has cleared nothing""", """    if (getenv("RTX_REGION_PRINT")) {
        unsigned long long r[6], it[2];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_region), sizeof r);
        (void)hipMemcpyFromSymbol(it, HIP_SYMBOL(g_iters), sizeof it);
        double tot = 0;
        for (int k = 0; k < 6; ++k) tot += (double)r[k];
        fprintf(stderr, "regions: taking tiles %.1f%%  loop and probes %.1f%%  before the mesh %.1f%%  mesh walk %.1f%%  planes, islands, hit record %.1f%%  "
                        "shading and bookkeeping %.1f%%  (%.3g wave clocks); %llu trips of the path loop, %.1f lanes casting\\n",
                100 * r[5] / tot, 100 * r[0] / tot, 100 * r[1] / tot, 100 * r[2] / tot, 100 * r[3] / tot, 100 * r[4] / tot, tot, it[0],
                it[0] ? (double)it[1] / it[0] : 0.0);
        memset(r, 0, sizeof r);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_region), r, sizeof r);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_iters), r, sizeof it);
    }
    ss.heads_flip = !ss.heads_flip;  // only now: a launch that failed has cleared nothing""")
    open(p, "w").write(s)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    host = [os.path.join(csrc, "host", f) for f in sorted(os.listdir(os.path.join(csrc, "host"))) if f.endswith(".cpp") and f != "rt_main.cpp"]
    cmd = [shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
           "-I" + os.path.join(R, "include"), "-ffp-contract=off", "-shared", "-Wl,-Bsymbolic", os.path.join(csrc, "rtx_hip.hip"),
           os.path.join(csrc, "rtx_group.cpp")] + host + ["-ldl", "-o", LIB]
    subprocess.check_call(cmd)
    shutil.rmtree(tmp, ignore_errors=True)
    print("built", LIB)


def run(workloads):
    os.environ["RTX_REGION_PRINT"] = "1"
    for w in workloads:
        print("==", w, flush=True)
        subprocess.call([sys.executable, os.path.join(R, "scripts", "counter_frames.py"), w, "4", LIB])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:] or ["4k", "planes4k", "bunny4k"])
