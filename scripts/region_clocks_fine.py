"""A finer version of region_clocks.py: wave clocks around sixteen pieces of the path loop of k_render (the parts of
castBegin / castEnd / pathConsume one by one), accumulated per wave in LDS by the first active lane at every probe.
The probes live in a temporary copy of the sources, never in the product.
usage: python scripts/region_clocks_fine.py build   (here, needs hipcc)
       python scripts/region_clocks_fine.py run [4k planes4k bunny4k]   (on a GPU box, after build)"""
import os
import shutil
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(R, "raytracing_amd", "librtx_hip_regions_fine.so")  # git-ignored like every built library

NAMES = ["loop", "filter", "spheres", "loose tris", "boxes", "mesh gate", "mesh walk", "water+islands", "hit record", "texel/checker",
         "shadow consume", "sky (miss)", "hit setup", "light aim", "node finish", "sky (in place)", "frame return", "tile take", "tile write"]
(LOOP, FILTER, SPHERES, TRIS, BOXES, GATE, MESH, PLANES, RECORD, TEXEL, SHCONS, SKY, HITSET, AIM, NODE, SKYIN, RETURN, TAKE,
 WRITE) = range(len(NAMES))
N = len(NAMES)


def patch(text, old, new, count=1):
    if text.count(old) < 1:
        raise SystemExit("region_clocks_fine: the sources have changed, this piece is gone:\n" + old)
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
    unsigned long long *acc = nullptr;  // this wave's accumulators in LDS
};
template <int K>
RTX_DEV void regionEnd(LaneCounters<false> &lc) {
    const long long now = __builtin_readcyclecounter();
    const unsigned long long m = __ballot(1);
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(0xffffffffu, __builtin_amdgcn_mbcnt_lo(0xffffffffu, 0u));
    if (lc.acc && lane == (unsigned)__builtin_ctzll(m)) lc.acc[K] += (unsigned long long)(now - lc.last);
    lc.last = now;
}
template <int K>
RTX_DEV void regionEnd(LaneCounters<true> &) {}
""")
    # ---- castBegin ----
    s = patch(s, """    if (!walked && !slabMaybe(q.object_bounds, q.object_bounds + 3, orig, inv)) return false;
""", """    if (!walked && !slabMaybe(q.object_bounds, q.object_bounds + 3, orig, inv)) return false;
    regionEnd<%d>(lc);
""" % FILTER)
    s = patch(s, """    // trianglesIntersect, :304-317
""", """    regionEnd<%d>(lc);
    // trianglesIntersect, :304-317
""" % SPHERES)
    s = patch(s, """    // cubesIntersect, :382-397
""", """    regionEnd<%d>(lc);
    // cubesIntersect, :382-397
""" % TRIS)
    s = patch(s, """    cs.minDist = minDist, cs.kind = kind, cs.index = index;
    // modelsIntersect, :350-380
    return s.n_mesh_tris > 0 && !(minDist < t_exit) && meshGate<STATS>(s, orig, dir, inv, lc);""", """    cs.minDist = minDist, cs.kind = kind, cs.index = index;
    regionEnd<%d>(lc);
    const bool need_walk = s.n_mesh_tris > 0 && !(minDist < t_exit) && meshGate<STATS>(s, orig, dir, inv, lc);
    regionEnd<%d>(lc);
    return need_walk;""" % (BOXES, GATE))
    # ---- castEnd ----
    s = patch(s, """    min_dist = minDist;
    if (shadow) return minDist < s.far_clip;
""", """    min_dist = minDist;
    regionEnd<%d>(lc);
    if (shadow) return minDist < s.far_clip;
""" % PLANES)
    s = patch(s, """    h.m = materialAt<WIDE>(s, q, mat);
    h.mat_index = mat;
""", """    h.m = materialAt<WIDE>(s, q, mat);
    h.mat_index = mat;
    regionEnd<%d>(lc);
""" % RECORD)
    s = patch(s, """        h.m.dr = c[0], h.m.dg = c[1], h.m.db = c[2];
    }
    return minDist < s.far_clip;""", """        h.m.dr = c[0], h.m.dg = c[1], h.m.db = c[2];
    }
    regionEnd<%d>(lc);
    return minDist < s.far_clip;""" % TEXEL)
    s = patch(s, """    CastState cs;
    if (castBegin<STATS, WIDE>(s, q, orig, dir, t_exit, cs, stack, stride, lc)) castMesh<STATS>(s, orig, dir, t_exit, cs, stack, stride, lc);
    return castEnd<STATS, WIDE>(s, q, orig, dir, shadow, t_exit, cs, h, min_dist, lc);""", """    CastState cs;
    regionEnd<%d>(lc);
    const bool need = castBegin<STATS, WIDE>(s, q, orig, dir, t_exit, cs, stack, stride, lc);
    regionEnd<%d>(lc);
    if (need) castMesh<STATS>(s, orig, dir, t_exit, cs, stack, stride, lc);
    regionEnd<%d>(lc);
    const bool r = castEnd<STATS, WIDE>(s, q, orig, dir, shadow, t_exit, cs, h, min_dist, lc);
    regionEnd<%d>(lc);
    return r;""" % (LOOP, FILTER, MESH, RECORD))
    # ---- pathConsume ----
    s = patch(s, """    bool returning = false;          // `color` holds the finished value of a node
    if (shadow && quad_j < 0) {""", """    bool returning = false;          // `color` holds the finished value of a node
    regionEnd<%d>(lc);
    if (shadow && quad_j < 0) {""" % LOOP)
    s = patch(s, """            pathAddLight(q, st, st.dir, st.ndl, lightAt<WIDE>(s, q, st.light).w);
        }
        ++st.light;
""", """            pathAddLight(q, st, st.dir, st.ndl, lightAt<WIDE>(s, q, st.light).w);
        }
        ++st.light;
        regionEnd<%d>(lc);
""" % SHCONS)
    s = patch(s, """        st.light += 4;
    } else if (!hit) {""", """        st.light += 4;
        regionEnd<%d>(lc);
    } else if (!hit) {""" % SHCONS)
    s = patch(s, """        color = skyColor(s, st.dir);
        returning = true;
    } else {
        st.point = h.point, st.N = h.N, st.m = h.m, st.view = st.dir;
        st.Ld = 0.0f, st.Ls = 0.0f;
        st.light = 0;
    }""", """        color = skyColor(s, st.dir);
        returning = true;
        regionEnd<%d>(lc);
    } else {
        st.point = h.point, st.N = h.N, st.m = h.m, st.view = st.dir;
        st.Ld = 0.0f, st.Ls = 0.0f;
        st.light = 0;
        regionEnd<%d>(lc);
    }""" % (SKY, HITSET))
    s = patch(s, """                pathAimShadowRay(s, st, a);
                return false;
            }""", """                pathAimShadowRay(s, st, a);
                regionEnd<%d>(lc);
                return false;
            }""" % AIM)
    s = patch(s, """            if (mask) {
                st.cast_mask = mask;
                return false;
            }""", """            if (mask) {
                st.cast_mask = mask;
                regionEnd<%d>(lc);
                return false;
            }""" % AIM)
    s = patch(s, """        // ---- all lights done: the node's own colour and its children, :96-105 ----
        st.light = -1, st.t_exit = -1.0f;""", """        // ---- all lights done: the node's own colour and its children, :96-105 ----
        regionEnd<%d>(lc);
        st.light = -1, st.t_exit = -1.0f;""" % AIM)
    s = patch(s, """            if (depth == 1) {
                // the children have depth 0: traceRay returns the sky colour of their direction at
                // once (src/Scene.cpp:87-88), so they are evaluated in place
                V3 c = base;""", """            if (depth == 1) {
                // the children have depth 0: traceRay returns the sky colour of their direction at
                // once (src/Scene.cpp:87-88), so they are evaluated in place
                regionEnd<%d>(lc);
                V3 c = base;""" % NODE)
    s = patch(s, """                color = c;
            } else {
                FrameHead f;""", """                color = c;
                regionEnd<%d>(lc);
            } else {
                FrameHead f;""" % SKYIN)
    s = patch(s, """                st.depth = depth - 1;
                return false;
            }
        }
    }
    // ---- return `color` to the waiting frames ----""", """                st.depth = depth - 1;
                regionEnd<%d>(lc);
                return false;
            }
        }
        regionEnd<%d>(lc);
    }
    // ---- return `color` to the waiting frames ----""" % (NODE, NODE))
    s = patch(s, """                st.depth = (f.state >> 8) - 1;
                return false;
            }""", """                st.depth = (f.state >> 8) - 1;
                regionEnd<%d>(lc);
                return false;
            }""" % RETURN)
    s = patch(s, """        --st.sp;
    }
    return true;
}""", """        --st.sp;
    }
    regionEnd<%d>(lc);
    return true;
}""" % RETURN)
    open(p, "w").write(s)

    p = os.path.join(csrc, "rtx_hip.hip")
    s = open(p).read()
    s = patch(s, """template <bool STATS>
RTX_DEV void flushCounters(const LaneCounters<STATS> &lc, Counters *out) {
    if constexpr (STATS) {""", """__device__ unsigned long long g_region[%d];
template <bool STATS>
RTX_DEV void flushCounters(const LaneCounters<STATS> &lc, Counters *out) {
    if constexpr (!STATS) {
        if ((threadIdx.x & 63) == 0 && lc.acc)
            for (int k = 0; k < %d; ++k) atomicAdd(&g_region[k], lc.acc[k]);
    }
    if constexpr (STATS) {""" % (N, N))
    s = patch(s, """    LaneCounters<STATS> lc;
    PathState st;
    FrameSpill spill;""", """    LaneCounters<STATS> lc;
    __shared__ unsigned long long s_regions[kBlock / 64][%d];
    if constexpr (!STATS) {
        if ((threadIdx.x & 63u) < %du) s_regions[threadIdx.x >> 6][threadIdx.x & 63u] = 0ull;
        lc.acc = s_regions[threadIdx.x >> 6];
        lc.last = __builtin_readcyclecounter();
    }
    PathState st;
    FrameSpill spill;""" % (N, N))
    s = patch(s, """        // ---- take a tile ----
        unsigned tile = 0xffffffffu;""", """        // ---- take a tile ----
        regionEnd<%d>(lc);
        unsigned tile = 0xffffffffu;""" % WRITE)
    s = patch(s, """        if (tile == 0xffffffffu) break;  // every queue is empty""", """        regionEnd<%d>(lc);
        if (tile == 0xffffffffu) break;  // every queue is empty""" % TAKE)
    s = patch(s, """        // what the tile cost (100 MHz ticks): the next frame of the same view starts the expensive ones first""", """        regionEnd<%d>(lc);
        // what the tile cost (100 MHz ticks): the next frame of the same view starts the expensive ones first""" % LOOP)
    names_c = ", ".join('"%s"' % n for n in NAMES)
    s = patch(s, """    ss.heads_flip = !ss.heads_flip;  // only now: a launch that failed has cleared nothing""", """    if (getenv("RTX_REGION_PRINT")) {
        static const char *names[%d] = {%s};
        unsigned long long r[%d];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpyFromSymbol(r, HIP_SYMBOL(g_region), sizeof r);
        double tot = 0;
        for (int k = 0; k < %d; ++k) tot += (double)r[k];
        fprintf(stderr, "regions (%%.3g wave clocks):", tot);
        for (int k = 0; k < %d; ++k) fprintf(stderr, "  %%s %%.1f%%%%", names[k], 100 * r[k] / tot);
        fprintf(stderr, "\\n");
        memset(r, 0, sizeof r);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_region), r, sizeof r);
    }
    ss.heads_flip = !ss.heads_flip;  // only now: a launch that failed has cleared nothing""" % (N, names_c, N, N, N))
    open(p, "w").write(s)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    host = [os.path.join(csrc, "host", f) for f in sorted(os.listdir(os.path.join(csrc, "host"))) if f.endswith(".cpp") and f != "rt_main.cpp"]
    cmd = [shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
           "-fno-slp-vectorize", "-I" + os.path.join(R, "include"), "-ffp-contract=off", "-shared", "-Wl,-Bsymbolic",
           os.path.join(csrc, "rtx_hip.hip"), os.path.join(csrc, "rtx_group.cpp")] + host + ["-ldl", "-o", LIB]
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
