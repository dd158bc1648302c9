// GPU construction of the mesh tree by the surface-area heuristic, top-down and binned: the algorithm of the host
// builder in sah_build.hpp (32 bins over the centroid bounds per axis, cost = area x count on either side, halving
// where no plane separates anything or the depth bound says so) run level by level, ONE WAVE PER NODE: the wave
// reduces the node's centroid bounds, fills the bins of all three axes in LDS with atomics, sweeps them, partitions
// the node's triangle references (ballot + mbcnt, from one array into the other), and writes the finished
// 64-byte node -- both children's boxes and ids -- before the children exist.  A level is one launch; the next
// level's nodes are appended to a queue with one atomic each.  The top levels are a few long waves (the root's
// wave walks every triangle three times), the bottom levels thousands of short ones; a 4 968-triangle mesh takes
// about twenty launches of a few microseconds, a million triangles a few tens of milliseconds.
//
// Replaces, like the other builders, the reference's 7x7x7 vertex-binned grid (BoundingBox::init / initTriangles,
// src/SceneObjects.cpp:176-209).  The tree only steers the walk.
#pragma once
#include <hip/hip_runtime.h>
#include "rtx_device.hpp"

namespace rtx {

constexpr int kSahBins = 32;

struct SahGpuRef {  // a triangle reference: padded box, centroid, triangle number (10 words)
    float lo[3], hi[3], c[3];
    int tri;
};
struct SahGpuTask {  // a node to build: its id and its range of references
    int id, begin, end, depth;
};

__device__ __forceinline__ unsigned sahOrdered(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sahUnordered(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__device__ __forceinline__ float sahWaveMin(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float sahWaveMax(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float sahHalfArea(const float lo[3], const float hi[3]) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx < 0 ? 0.0f : dx * dy + dy * dz + dz * dx;
}

__global__ void k_sah_refs(const float *__restrict__ verts, const int *__restrict__ idx, int nt, float pad, SahGpuRef *__restrict__ refs) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nt) return;
    SahGpuRef r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = verts[3 * (size_t)idx[3 * t] + k], b = verts[3 * (size_t)idx[3 * t + 1] + k], c = verts[3 * (size_t)idx[3 * t + 2] + k];
        const float lo = fminf(a, fminf(b, c)), hi = fmaxf(a, fmaxf(b, c));
        r.c[k] = 0.5f * (lo + hi);
        r.lo[k] = lo - pad, r.hi[k] = hi + pad;
    }
    r.tri = t;
    refs[t] = r;
}

// counters: [0] nodes allocated, [1] tasks queued for the next level, [2] deepest leaf
__global__ __launch_bounds__(256) void k_sah_level(const SahGpuTask *__restrict__ tasks, int n_tasks, const SahGpuRef *__restrict__ in,
                                                   SahGpuRef *__restrict__ out, SahGpuTask *__restrict__ next, int *__restrict__ counters,
                                                   BvhNode *__restrict__ nodes, int depth_limit) {
    // per wave: bins of the three axes -- box (6 ordered words) and count each
    __shared__ unsigned s_bins[4][3][kSahBins][7];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + (int)wave;
    if (task >= n_tasks) return;
    const SahGpuTask tk = tasks[task];
    const int b = tk.begin, e = tk.end, n = e - b;
    unsigned(*bins)[kSahBins][7] = s_bins[wave];
    // ---- centroid bounds ----
    float clo[3] = {kFltMax, kFltMax, kFltMax}, chi[3] = {-kFltMax, -kFltMax, -kFltMax};
    for (int i = b + (int)lane; i < e; i += 64)
#pragma unroll
        for (int k = 0; k < 3; ++k) clo[k] = fminf(clo[k], in[i].c[k]), chi[k] = fmaxf(chi[k], in[i].c[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) clo[k] = sahWaveMin(clo[k]), chi[k] = sahWaveMax(chi[k]);
    // ---- the plane: binned surface-area heuristic, unless the remaining levels are needed for plain halving ----
    int levels_needed = 0;
    while ((1ll << levels_needed) < (long long)n) ++levels_needed;
    const bool forced_median = tk.depth + levels_needed >= depth_limit;
    int best_axis = -1, best_bin = -1;
    float scale[3] = {0.0f, 0.0f, 0.0f};
    if (!forced_median && n > 2) {
        for (unsigned w = lane; w < 3u * kSahBins * 7u; w += 64u) {
            const unsigned slot = w % 7u;
            (&bins[0][0][0])[w] = slot < 3u ? sahOrdered(kFltMax) : (slot < 6u ? sahOrdered(-kFltMax) : 0u);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float ext = chi[k] - clo[k];
            scale[k] = ext > 0.0f ? kSahBins / ext : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = b + (int)lane; i < e; i += 64) {
            const SahGpuRef r = in[i];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (!(scale[k] > 0.0f)) continue;
                int bin = (int)((r.c[k] - clo[k]) * scale[k]);
                bin = bin < 0 ? 0 : (bin > kSahBins - 1 ? kSahBins - 1 : bin);
                unsigned *s = bins[k][bin];
                atomicMin(&s[0], sahOrdered(r.lo[0])), atomicMin(&s[1], sahOrdered(r.lo[1])), atomicMin(&s[2], sahOrdered(r.lo[2]));
                atomicMax(&s[3], sahOrdered(r.hi[0])), atomicMax(&s[4], sahOrdered(r.hi[1])), atomicMax(&s[5], sahOrdered(r.hi[2]));
                atomicAdd(&s[6], 1u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // lanes 0..2 sweep one axis each; the cheapest plane wins, the lower axis and bin on equal cost
        float cost = kFltMax;
        int bin_of = -1;
        const float my_scale = lane == 0u ? scale[0] : (lane == 1u ? scale[1] : scale[2]);
        if (lane < 3u && my_scale > 0.0f) {
            float right_area[kSahBins];
            float lo[3] = {kFltMax, kFltMax, kFltMax}, hi[3] = {-kFltMax, -kFltMax, -kFltMax};
            for (int k = kSahBins - 1; k > 0; --k) {
                const unsigned *s = bins[lane][k];
#pragma unroll
                for (int a = 0; a < 3; ++a) lo[a] = fminf(lo[a], sahUnordered(s[a])), hi[a] = fmaxf(hi[a], sahUnordered(s[3 + a]));
                right_area[k] = sahHalfArea(lo, hi);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) lo[a] = kFltMax, hi[a] = -kFltMax;
            int left_n = 0;
            for (int k = 0; k + 1 < kSahBins; ++k) {
                const unsigned *s = bins[lane][k];
#pragma unroll
                for (int a = 0; a < 3; ++a) lo[a] = fminf(lo[a], sahUnordered(s[a])), hi[a] = fmaxf(hi[a], sahUnordered(s[3 + a]));
                left_n += (int)s[6];
                if (left_n == 0 || left_n == n) continue;
                const float c = sahHalfArea(lo, hi) * (float)left_n + right_area[k + 1] * (float)(n - left_n);
                if (c < cost) cost = c, bin_of = k;
            }
        }
        const float c0 = __shfl(cost, 0), c1 = __shfl(cost, 1), c2 = __shfl(cost, 2);
        const int b0 = __shfl(bin_of, 0), b1 = __shfl(bin_of, 1), b2 = __shfl(bin_of, 2);
        float best = kFltMax;
        if (b0 >= 0 && c0 < best) best = c0, best_axis = 0, best_bin = b0;
        if (b1 >= 0 && c1 < best) best = c1, best_axis = 1, best_bin = b1;
        if (b2 >= 0 && c2 < best) best = c2, best_axis = 2, best_bin = b2;
    }
    // ---- partition [b, e) from `in` into `out`: the left side first, both in their old order.  Without a plane
    // the range is simply halved as it stands. ----
    const float cut_lo = best_axis == 0 ? clo[0] : (best_axis == 1 ? clo[1] : clo[2]);
    const float cut_scale = best_axis == 0 ? scale[0] : (best_axis == 1 ? scale[1] : scale[2]);
    auto goesLeft = [&](const SahGpuRef &r, int i) {
        if (best_axis < 0) return i < b + n / 2;
        const float c = best_axis == 0 ? r.c[0] : (best_axis == 1 ? r.c[1] : r.c[2]);
        int bin = (int)((c - cut_lo) * cut_scale);
        bin = bin < 0 ? 0 : (bin > kSahBins - 1 ? kSahBins - 1 : bin);
        return bin <= best_bin;
    };
    int n_left = 0;
    for (int i0 = b; i0 < e; i0 += 64) {
        const int i = i0 + (int)lane;
        const bool l = i < e && goesLeft(in[i], i);
        n_left += __popcll(__ballot(l));
    }
    if (n_left == 0 || n_left == n) {  // (cannot happen with a plane from the sweep; guards the arithmetic)
        best_axis = -1;
        n_left = n / 2;
    }
    const int mid = b + n_left;
    float box[2][6];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int a = 0; a < 3; ++a) box[s][a] = kFltMax, box[s][3 + a] = -kFltMax;
    int leaf_tri[2] = {-1, -1};  // the triangle of a side that holds just one
    int done_left = 0, done_right = 0;
    for (int i0 = b; i0 < e; i0 += 64) {
        const int i = i0 + (int)lane;
        const bool live = i < e;
        SahGpuRef r;
        bool l = false;
        if (live) r = in[i], l = goesLeft(r, i);
        const unsigned long long ml = __ballot(live && l), mr = __ballot(live && !l);
        if (live) {
            const int side = l ? 0 : 1;
            const int at = l ? b + done_left + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ml >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ml, 0u))
                             : mid + done_right + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mr >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mr, 0u));
            out[at] = r;
#pragma unroll
            for (int a = 0; a < 3; ++a) box[side][a] = fminf(box[side][a], r.lo[a]), box[side][3 + a] = fmaxf(box[side][3 + a], r.hi[a]);
            if (at == b) leaf_tri[0] = r.tri;
            if (at == mid) leaf_tri[1] = r.tri;
        }
        done_left += __popcll(ml), done_right += __popcll(mr);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int a = 0; a < 3; ++a) box[s][a] = sahWaveMin(box[s][a]), box[s][3 + a] = sahWaveMax(box[s][3 + a]);
        for (int off = 32; off > 0; off >>= 1) leaf_tri[s] = max(leaf_tri[s], __shfl_xor(leaf_tri[s], off));
    }
    // ---- the node, and its children's tasks ----
    if (lane == 0) {
        int child[2];
        const int lo_[2] = {b, mid}, hi_[2] = {mid, e};
        for (int s = 0; s < 2; ++s) {
            if (hi_[s] - lo_[s] == 1) {
                child[s] = ~leaf_tri[s];
                atomicMax(&counters[2], tk.depth);
            } else {
                child[s] = atomicAdd(&counters[0], 1);
                next[atomicAdd(&counters[1], 1)] = SahGpuTask{child[s], lo_[s], hi_[s], tk.depth + 1};
            }
        }
        BvhNode nd;
        nd.lminx = box[0][0], nd.lminy = box[0][1], nd.lminz = box[0][2], nd.lmaxx = box[0][3], nd.lmaxy = box[0][4], nd.lmaxz = box[0][5];
        nd.rminx = box[1][0], nd.rminy = box[1][1], nd.rminz = box[1][2], nd.rmaxx = box[1][3], nd.rmaxy = box[1][4], nd.rmaxz = box[1][5];
        nd.left = child[0], nd.right = child[1], nd.pad0 = 0, nd.pad1 = 0;
        nodes[tk.id] = nd;
    }
}

// Builds the tree over nt >= 2 triangles (device arrays verts, idx) into nodes[0 .. nt-2], root = node 0.
// Synchronises the stream once per level.
inline hipError_t sahGpuBuild(const float *d_verts, const int *d_idx, int nt, float pad, int depth_limit, hipStream_t stream, BvhNode *nodes,
                              int *max_depth_out) {
    hipError_t err = hipSuccess;
    SahGpuRef *refs[2] = {nullptr, nullptr};
    SahGpuTask *tasks[2] = {nullptr, nullptr};
    int *counters = nullptr;
    int host[3] = {1, 0, 0};
    int n_tasks = 1, cur = 0;
    const SahGpuTask root{0, 0, nt, 1};
#define SAH_TRY(expr)                      \
    do {                                   \
        err = (expr);                      \
        if (err != hipSuccess) goto done;  \
    } while (0)
    SAH_TRY(hipMalloc(&refs[0], sizeof(SahGpuRef) * (size_t)nt));
    SAH_TRY(hipMalloc(&refs[1], sizeof(SahGpuRef) * (size_t)nt));
    SAH_TRY(hipMalloc(&tasks[0], sizeof(SahGpuTask) * (size_t)nt));
    SAH_TRY(hipMalloc(&tasks[1], sizeof(SahGpuTask) * (size_t)nt));
    SAH_TRY(hipMalloc(&counters, sizeof(int) * 3));
    SAH_TRY(hipMemcpyAsync(counters, host, sizeof host, hipMemcpyHostToDevice, stream));
    SAH_TRY(hipMemcpyAsync(tasks[0], &root, sizeof root, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_sah_refs, dim3((nt + 255) / 256), dim3(256), 0, stream, d_verts, d_idx, nt, pad, refs[0]);
    for (int level = 0; n_tasks > 0; ++level) {
        if (level > depth_limit + 1) {  // (the depth bound makes this unreachable)
            err = hipErrorUnknown;
            goto done;
        }
        hipLaunchKernelGGL(k_sah_level, dim3((n_tasks + 3) / 4), dim3(256), 0, stream, tasks[cur], n_tasks, refs[cur], refs[cur ^ 1], tasks[cur ^ 1],
                           counters, nodes, depth_limit);
        SAH_TRY(hipGetLastError());
        SAH_TRY(hipMemcpyAsync(host, counters, sizeof host, hipMemcpyDeviceToHost, stream));
        SAH_TRY(hipStreamSynchronize(stream));
        n_tasks = host[1];
        SAH_TRY(hipMemsetAsync(counters + 1, 0, sizeof(int), stream));
        cur ^= 1;
    }
    if (host[0] != nt - 1) {
        err = hipErrorUnknown;
        goto done;
    }
    *max_depth_out = host[2];
done:
#undef SAH_TRY
    void *scratch[] = {refs[0], refs[1], tasks[0], tasks[1], counters};
    for (void *p : scratch) (void)hipFree(p);
    return err;
}

}  // namespace rtx
