"""When do the waves of k_render finish?  Builds a copy of the library in which every wave records the time at which
it leaves the kernel (100 MHz wall clock, relative to the first wave's start) and prints the distribution per frame:
the mean over waves against the last one is the share of the wave slots that idles in the frame's tail.
The probes live in a temporary copy of the sources, never in the product.
usage: python scripts/wave_ends.py build   (here, needs hipcc)
       python scripts/wave_ends.py run [4k planes4k bunny4k 8k]   (on a GPU box, after build)"""
import os
import shutil
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(R, "raytracing_amd", "librtx_hip_wave_ends.so")  # git-ignored like every built library


def patch(text, old, new):
    if text.count(old) < 1:
        raise SystemExit("wave_ends: the sources have changed, this piece is gone:\n" + old)
    return text.replace(old, new, 1)


def build():
    tmp = tempfile.mkdtemp(prefix="rtx_wave_ends_")
    os.makedirs(os.path.join(tmp, "raytracing_amd"))
    csrc = os.path.join(tmp, "raytracing_amd", "csrc")
    shutil.copytree(os.path.join(R, "raytracing_amd", "csrc"), csrc)
    os.symlink(os.path.join(R, "include"), os.path.join(tmp, "include"))
    p = os.path.join(csrc, "rtx_hip.hip")
    s = open(p).read()
    s = patch(s, """template <bool STATS>
RTX_DEV void flushCounters(const LaneCounters<STATS> &lc, Counters *out) {
    if constexpr (STATS) {""", """__device__ unsigned long long g_wave_start;  // earliest start
__device__ unsigned long long g_wave_sum, g_wave_max, g_wave_n;
__device__ unsigned g_wave_hist[3][512];  // start / finish times in 10 us steps after the first start (which the first waves estimate)
template <bool STATS>
RTX_DEV void flushCounters(const LaneCounters<STATS> &lc, Counters *out) {
    if constexpr (!STATS) {
        if ((threadIdx.x & 63) == 0) {
            const unsigned long long now = wall_clock64();
            atomicAdd(&g_wave_sum, now), atomicMax(&g_wave_max, now), atomicAdd(&g_wave_n, 1ull);
            const unsigned long long bin = (now - g_wave_start) / 1000ull;
            atomicAdd(&g_wave_hist[1][bin < 511ull ? bin : 511ull], 1u);
        }
    }
    if constexpr (STATS) {""")
    s = patch(s, """    LaneCounters<STATS> lc;
    PathState st;""", """    LaneCounters<STATS> lc;
    if constexpr (!STATS) {
        if ((threadIdx.x & 63) == 0) {
            const unsigned long long now = wall_clock64();
            const unsigned long long old = atomicMin(&g_wave_start, now);
            const unsigned long long bin = now > old ? (now - old) / 1000ull : 0ull;
            atomicAdd(&g_wave_hist[0][bin < 511ull ? bin : 511ull], 1u);
        }
    }
    PathState st;""")
    s = patch(s, """    ss.heads_flip = !ss.heads_flip;  // only now: a launch that failed has cleared nothing""", """    if (getenv("RTX_WAVE_ENDS")) {
        unsigned long long v[4];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(&v[0], HIP_SYMBOL(g_wave_start), 8);
        (void)hipMemcpyFromSymbol(&v[1], HIP_SYMBOL(g_wave_sum), 8);
        (void)hipMemcpyFromSymbol(&v[2], HIP_SYMBOL(g_wave_max), 8);
        (void)hipMemcpyFromSymbol(&v[3], HIP_SYMBOL(g_wave_n), 8);
        if (v[3]) {
            const double mean = ((double)v[1] / (double)v[3] - (double)v[0]) / 100.0, last = ((double)v[2] - (double)v[0]) / 100.0;
            fprintf(stderr, "waves %llu: mean finish %.1f us, last %.1f us, idle share of the slots %.1f %%\\n", v[3], mean, last,
                    100.0 * (1.0 - mean / last));
        }
        static unsigned hist[3][512];
        (void)hipMemcpyFromSymbol(hist, HIP_SYMBOL(g_wave_hist), sizeof hist);
        for (int k = 0; k < 3; ++k) {
            fprintf(stderr, k == 0 ? "  starts per 10 us:" : (k == 1 ? "  finishes per 10 us:" : "  own queue first found empty, per 10 us:"));
            int first = 0, last = 511;
            while (first < 511 && !hist[k][first]) ++first;
            while (last > 0 && !hist[k][last]) --last;
            fprintf(stderr, " [from %d us]", first * 10);
            for (int b = first; b <= last; ++b) fprintf(stderr, " %u", hist[k][b]);
            fprintf(stderr, "\\n");
        }
        memset(hist, 0, sizeof hist);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_hist), hist, sizeof hist);
        const unsigned long long big = ~0ull, zero = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_start), &big, 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_sum), &zero, 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_max), &zero, 8);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_n), &zero, 8);
    }
    ss.heads_flip = !ss.heads_flip;  // only now: a launch that failed has cleared nothing""")
    s = patch(s, """            // this queue is empty: every lane looks at one head, the wave moves to the next queue""", """            if (lane == 0) {
                if (!probe_dry) {
                    const unsigned long long bin = ((unsigned long long)wall_clock64() - g_wave_start) / 1000ull;
                    atomicAdd(&g_wave_hist[2][bin < 511ull ? bin : 511ull], 1u);
                }
                atomicAdd(&g_wave_moves, 1ull);
            }
            probe_dry = true;
            // this queue is empty: every lane looks at one head, the wave moves to the next queue""")
    s = patch(s, """    LaneCounters<STATS> lc;
    if constexpr (!STATS) {""", """    bool probe_dry = false;
    LaneCounters<STATS> lc;
    if constexpr (!STATS) {""")
    s = patch(s, """__device__ unsigned long long g_wave_sum,""", """__device__ unsigned long long g_wave_moves;
__device__ unsigned long long g_wave_sum,""")
    s = patch(s, """        static unsigned hist[3][512];""", """        unsigned long long moves = 0;
        (void)hipMemcpyFromSymbol(&moves, HIP_SYMBOL(g_wave_moves), 8);
        fprintf(stderr, "  queue found empty %llu times\\n", moves);
        moves = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_moves), &moves, 8);
        static unsigned hist[3][512];""")
    open(p, "w").write(s)
    host = [os.path.join(csrc, "host", f) for f in sorted(os.listdir(os.path.join(csrc, "host"))) if f.endswith(".cpp") and f != "rt_main.cpp"]
    cmd = [shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
           "-fno-slp-vectorize", "-I" + os.path.join(R, "include"), "-ffp-contract=off", "-shared", "-Wl,-Bsymbolic",
           os.path.join(csrc, "rtx_hip.hip"), os.path.join(csrc, "rtx_group.cpp")] + host + ["-ldl", "-o", LIB]
    subprocess.check_call(cmd)
    shutil.rmtree(tmp, ignore_errors=True)
    print("built", LIB)


def run(workloads):
    os.environ["RTX_WAVE_ENDS"] = "1"
    for w in workloads:
        print("==", w, flush=True)
        subprocess.call([sys.executable, os.path.join(R, "scripts", "counter_frames.py"), w, "5", LIB])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:] or ["4k", "planes4k", "bunny4k", "8k"])
