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

// The sweep over the filled bins, by a whole wave: lanes 0..2 take one axis each; the cheapest plane wins, the lower
// axis and bin on equal cost.  bins[axis][bin] = box (6 ordered words) and count.  best_axis stays -1 without a plane.
__device__ __forceinline__ void sahSweep(unsigned (*bins)[kSahBins][7], unsigned lane, int n, const float scale[3], int &best_axis, int &best_bin) {
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
// which side of the plane a reference goes to (the same arithmetic as the binning); without a plane the range is halved as it stands
__device__ __forceinline__ bool sahGoesLeft(const SahGpuRef &r, int i, int b, int n, int best_axis, int best_bin, float cut_lo, float cut_scale) {
    if (best_axis < 0) return i < b + n / 2;
    const float c = best_axis == 0 ? r.c[0] : (best_axis == 1 ? r.c[1] : r.c[2]);
    int bin = (int)((c - cut_lo) * cut_scale);
    bin = bin < 0 ? 0 : (bin > kSahBins - 1 ? kSahBins - 1 : bin);
    return bin <= best_bin;
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

// counters: [0] nodes allocated, [1] tasks queued for the next level, [2] deepest leaf ([3..5]: see k_sah_sort_tasks)
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
        sahSweep(bins, lane, n, scale, best_axis, best_bin);
    }
    // ---- partition [b, e) from `in` into `out`: the left side first, both in their old order.  Without a plane
    // the range is simply halved as it stands. ----
    const float cut_lo = best_axis == 0 ? clo[0] : (best_axis == 1 ? clo[1] : clo[2]);
    const float cut_scale = best_axis == 0 ? scale[0] : (best_axis == 1 ? scale[1] : scale[2]);
    auto goesLeft = [&](const SahGpuRef &r, int i) { return sahGoesLeft(r, i, b, n, best_axis, best_bin, cut_lo, cut_scale); };
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


// ---------------------------------------------------------------------------------------------------------------
// Large nodes.  One wave per node leaves the top of the tree to a handful of waves: the root's wave of a 1.27 M-triangle
// mesh walks every reference four times (40 ms), and a lopsided split keeps one long node per level for many levels.
// Nodes of more than kSahBigRefs references are therefore built by many workgroups, in chunks of kSahChunk references,
// one small launch per step: centroid bounds and bins by atomics (per-workgroup in LDS first), the sweep by one wave
// per node, then a stable partition -- lefts per chunk counted, scanned per node, scattered in order.  Same bounds,
// same bins, same sweep, same order of the references as the one-wave path: the tree does not depend on which path
// built a node.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSahBigRefs = 4096, kSahChunk = 2048, kSahBigThreads = 256;
struct SahBigNode {
    SahGpuTask task;
    int chunk_first, n_chunks;           // its chunks in the chunk arrays
    unsigned cb[6];                      // centroid bounds (ordered words: min xyz, max xyz)
    unsigned bins[3][kSahBins][7];
    unsigned child_box[2][6];            // ordered words
    int axis, bin, n_left;               // the plane (axis -1: halved as it stands) and how many references go left
    float cut_lo, cut_scale;
    int leaf_tri[2];
};

// sorts the level's tasks into small ones (one wave each) and large ones; counters: [3] small, [4] large, [5] chunks
__global__ void k_sah_sort_tasks(const SahGpuTask *__restrict__ tasks, int n_tasks, SahGpuTask *__restrict__ small, SahBigNode *__restrict__ big,
                                 int *__restrict__ chunk_node, int *__restrict__ counters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    const SahGpuTask tk = tasks[t];
    const int n = tk.end - tk.begin;
    if (n <= kSahBigRefs) {
        small[atomicAdd(&counters[3], 1)] = tk;
        return;
    }
    const int k = atomicAdd(&counters[4], 1), chunks = (n + kSahChunk - 1) / kSahChunk;
    const int first = atomicAdd(&counters[5], chunks);
    SahBigNode &nd = big[k];
    nd.task = tk, nd.chunk_first = first, nd.n_chunks = chunks;
    for (int a = 0; a < 3; ++a) nd.cb[a] = sahOrdered(kFltMax), nd.cb[3 + a] = sahOrdered(-kFltMax);
    for (int w = 0; w < 3 * kSahBins * 7; ++w) {
        const int slot = w % 7;
        (&nd.bins[0][0][0])[w] = slot < 3 ? sahOrdered(kFltMax) : (slot < 6 ? sahOrdered(-kFltMax) : 0u);
    }
    for (int s = 0; s < 2; ++s)
        for (int a = 0; a < 3; ++a) nd.child_box[s][a] = sahOrdered(kFltMax), nd.child_box[s][3 + a] = sahOrdered(-kFltMax);
    nd.axis = -1, nd.bin = -1, nd.n_left = 0, nd.cut_lo = 0.0f, nd.cut_scale = 0.0f, nd.leaf_tri[0] = nd.leaf_tri[1] = -1;
    for (int j = 0; j < chunks; ++j) chunk_node[first + j] = k;
}

// the references of chunk `c`: [i0, i1) within its node's range
__device__ __forceinline__ void sahChunkRange(const SahBigNode &nd, int c, int &i0, int &i1) {
    i0 = nd.task.begin + (c - nd.chunk_first) * kSahChunk;
    i1 = min(i0 + kSahChunk, nd.task.end);
}

__global__ __launch_bounds__(kSahBigThreads) void k_sah_big_bounds(const int *__restrict__ chunk_node, SahBigNode *__restrict__ big, const SahGpuRef *__restrict__ in) {
    SahBigNode &nd = big[chunk_node[blockIdx.x]];
    int i0, i1;
    sahChunkRange(nd, (int)blockIdx.x, i0, i1);
    float clo[3] = {kFltMax, kFltMax, kFltMax}, chi[3] = {-kFltMax, -kFltMax, -kFltMax};
    for (int i = i0 + (int)threadIdx.x; i < i1; i += kSahBigThreads)
#pragma unroll
        for (int k = 0; k < 3; ++k) clo[k] = fminf(clo[k], in[i].c[k]), chi[k] = fmaxf(chi[k], in[i].c[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        clo[k] = sahWaveMin(clo[k]), chi[k] = sahWaveMax(chi[k]);
        if ((threadIdx.x & 63u) == 0u) atomicMin(&nd.cb[k], sahOrdered(clo[k])), atomicMax(&nd.cb[3 + k], sahOrdered(chi[k]));
    }
}

__global__ __launch_bounds__(kSahBigThreads) void k_sah_big_bins(const int *__restrict__ chunk_node, SahBigNode *__restrict__ big, const SahGpuRef *__restrict__ in,
                                                                  int depth_limit) {
    __shared__ unsigned s_bins[3][kSahBins][7];
    SahBigNode &nd = big[chunk_node[blockIdx.x]];
    const int n = nd.task.end - nd.task.begin;
    int levels_needed = 0;
    while ((1ll << levels_needed) < (long long)n) ++levels_needed;
    if (nd.task.depth + levels_needed >= depth_limit) return;  // halved as it stands: no bins needed
    int i0, i1;
    sahChunkRange(nd, (int)blockIdx.x, i0, i1);
    for (unsigned w = threadIdx.x; w < 3u * kSahBins * 7u; w += kSahBigThreads) {
        const unsigned slot = w % 7u;
        (&s_bins[0][0][0])[w] = slot < 3u ? sahOrdered(kFltMax) : (slot < 6u ? sahOrdered(-kFltMax) : 0u);
    }
    float clo[3], scale[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        clo[k] = sahUnordered(nd.cb[k]);
        const float ext = sahUnordered(nd.cb[3 + k]) - clo[k];
        scale[k] = ext > 0.0f ? kSahBins / ext : 0.0f;
    }
    __syncthreads();
    for (int i = i0 + (int)threadIdx.x; i < i1; i += kSahBigThreads) {
        const SahGpuRef r = in[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!(scale[k] > 0.0f)) continue;
            int bin = (int)((r.c[k] - clo[k]) * scale[k]);
            bin = bin < 0 ? 0 : (bin > kSahBins - 1 ? kSahBins - 1 : bin);
            unsigned *s = s_bins[k][bin];
            atomicMin(&s[0], sahOrdered(r.lo[0])), atomicMin(&s[1], sahOrdered(r.lo[1])), atomicMin(&s[2], sahOrdered(r.lo[2]));
            atomicMax(&s[3], sahOrdered(r.hi[0])), atomicMax(&s[4], sahOrdered(r.hi[1])), atomicMax(&s[5], sahOrdered(r.hi[2]));
            atomicAdd(&s[6], 1u);
        }
    }
    __syncthreads();
    for (unsigned w = threadIdx.x; w < 3u * kSahBins * 7u; w += kSahBigThreads) {
        const unsigned slot = w % 7u, v = (&s_bins[0][0][0])[w];
        unsigned *g = &(&nd.bins[0][0][0])[w];
        if (slot < 3u) atomicMin(g, v);
        else if (slot < 6u) atomicMax(g, v);
        else if (v) atomicAdd(g, v);
    }
}

// one wave per large node: the plane
__global__ __launch_bounds__(64) void k_sah_big_plane(SahBigNode *__restrict__ big, int depth_limit) {
    __shared__ unsigned s_bins[3][kSahBins][7];
    SahBigNode &nd = big[blockIdx.x];
    const unsigned lane = threadIdx.x;
    const int n = nd.task.end - nd.task.begin;
    int levels_needed = 0;
    while ((1ll << levels_needed) < (long long)n) ++levels_needed;
    int best_axis = -1, best_bin = -1;
    float clo[3], scale[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        clo[k] = sahUnordered(nd.cb[k]);
        const float ext = sahUnordered(nd.cb[3 + k]) - clo[k];
        scale[k] = ext > 0.0f ? kSahBins / ext : 0.0f;
    }
    if (nd.task.depth + levels_needed < depth_limit) {
        for (unsigned w = lane; w < 3u * kSahBins * 7u; w += 64u) (&s_bins[0][0][0])[w] = (&nd.bins[0][0][0])[w];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        sahSweep(s_bins, lane, n, scale, best_axis, best_bin);
    }
    if (lane == 0) {
        nd.axis = best_axis, nd.bin = best_bin;
        nd.cut_lo = best_axis == 0 ? clo[0] : (best_axis == 1 ? clo[1] : clo[2]);
        nd.cut_scale = best_axis == 0 ? scale[0] : (best_axis == 1 ? scale[1] : scale[2]);
    }
}

// lefts per chunk
__global__ __launch_bounds__(kSahBigThreads) void k_sah_big_count(const int *__restrict__ chunk_node, const SahBigNode *__restrict__ big,
                                                                   const SahGpuRef *__restrict__ in, int *__restrict__ chunk_left) {
    __shared__ int s_n[kSahBigThreads / 64];
    const SahBigNode &nd = big[chunk_node[blockIdx.x]];
    const int b = nd.task.begin, n = nd.task.end - b;
    int i0, i1;
    sahChunkRange(nd, (int)blockIdx.x, i0, i1);
    int mine = 0;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += kSahBigThreads) mine += sahGoesLeft(in[i], i, b, n, nd.axis, nd.bin, nd.cut_lo, nd.cut_scale) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63u) == 0u) s_n[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < kSahBigThreads / 64; ++w) total += s_n[w];
        chunk_left[blockIdx.x] = total;
    }
}

// one wave per large node: lefts before each of its chunks; a plane that separates nothing becomes a halving
__global__ __launch_bounds__(64) void k_sah_big_scan(SahBigNode *__restrict__ big, int *__restrict__ chunk_left) {
    SahBigNode &nd = big[blockIdx.x];
    const int lane = (int)threadIdx.x, n = nd.task.end - nd.task.begin;
    int total = 0;
    for (int c0 = 0; c0 < nd.n_chunks; c0 += 64) total += (c0 + lane < nd.n_chunks) ? chunk_left[nd.chunk_first + c0 + lane] : 0;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    const bool halve = total == 0 || total == n;  // (cannot happen with a plane from the sweep; guards the arithmetic)
    int running = 0;
    for (int c0 = 0; c0 < nd.n_chunks; c0 += 64) {
        const int c = c0 + lane;
        int v = 0;
        if (c < nd.n_chunks) v = halve ? max(0, min(kSahChunk, n / 2 - c * kSahChunk)) : chunk_left[nd.chunk_first + c];
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (c < nd.n_chunks) chunk_left[nd.chunk_first + c] = running + incl - v;
        running += __shfl(incl, 63);
    }
    if (lane == 0) {
        if (halve) nd.axis = -1;
        nd.n_left = halve ? n / 2 : total;
    }
}

// the stable partition of a chunk: lefts to begin + (lefts before), rights to mid + (rights before), in their old order
__global__ __launch_bounds__(kSahBigThreads) void k_sah_big_scatter(const int *__restrict__ chunk_node, SahBigNode *__restrict__ big, const SahGpuRef *__restrict__ in,
                                                                     SahGpuRef *__restrict__ out, const int *__restrict__ chunk_left) {
    __shared__ int s_cnt[2][kSahBigThreads / 64];
    SahBigNode &nd = big[chunk_node[blockIdx.x]];
    const int b = nd.task.begin, n = nd.task.end - b, mid = b + nd.n_left;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    int i0, i1;
    sahChunkRange(nd, (int)blockIdx.x, i0, i1);
    int done_left = chunk_left[blockIdx.x], done_right = (i0 - b) - done_left;
    float box[2][6];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int a = 0; a < 3; ++a) box[s][a] = kFltMax, box[s][3 + a] = -kFltMax;
    for (int base = i0; base < i1; base += kSahBigThreads) {
        const int i = base + (int)threadIdx.x;
        const bool live = i < i1;
        SahGpuRef r;
        bool l = false;
        if (live) r = in[i], l = sahGoesLeft(r, i, b, n, nd.axis, nd.bin, nd.cut_lo, nd.cut_scale);
        const unsigned long long ml = __ballot(live && l), mr = __ballot(live && !l);
        if (lane == 0) s_cnt[0][wave] = __popcll(ml), s_cnt[1][wave] = __popcll(mr);
        __syncthreads();
        int before_l = 0, before_r = 0, all_l = 0, all_r = 0;
#pragma unroll
        for (unsigned w = 0; w < kSahBigThreads / 64; ++w) {
            before_l += w < wave ? s_cnt[0][w] : 0, before_r += w < wave ? s_cnt[1][w] : 0;
            all_l += s_cnt[0][w], all_r += s_cnt[1][w];
        }
        if (live) {
            const int side = l ? 0 : 1;
            const int at = l ? b + done_left + before_l + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ml >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ml, 0u))
                             : mid + done_right + before_r + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mr >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mr, 0u));
            out[at] = r;
#pragma unroll
            for (int a = 0; a < 3; ++a) box[side][a] = fminf(box[side][a], r.lo[a]), box[side][3 + a] = fmaxf(box[side][3 + a], r.hi[a]);
            if (at == b) nd.leaf_tri[0] = r.tri;
            if (at == mid) nd.leaf_tri[1] = r.tri;
        }
        done_left += all_l, done_right += all_r;
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = sahWaveMin(box[s][a]), hi = sahWaveMax(box[s][3 + a]);
            if (lane == 0) atomicMin(&nd.child_box[s][a], sahOrdered(lo)), atomicMax(&nd.child_box[s][3 + a], sahOrdered(hi));
        }
}

// one thread per large node: the node and its children's tasks (as the last step of k_sah_level)
__global__ void k_sah_big_finish(const SahBigNode *__restrict__ big, int n_big, SahGpuTask *__restrict__ next, int *__restrict__ counters,
                                 BvhNode *__restrict__ nodes) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_big) return;
    const SahBigNode &nd = big[t];
    const SahGpuTask tk = nd.task;
    const int b = tk.begin, e = tk.end, mid = b + nd.n_left;
    int child[2];
    const int lo_[2] = {b, mid}, hi_[2] = {mid, e};
    for (int s = 0; s < 2; ++s) {
        if (hi_[s] - lo_[s] == 1) {
            child[s] = ~nd.leaf_tri[s];
            atomicMax(&counters[2], tk.depth);
        } else {
            child[s] = atomicAdd(&counters[0], 1);
            next[atomicAdd(&counters[1], 1)] = SahGpuTask{child[s], lo_[s], hi_[s], tk.depth + 1};
        }
    }
    float box[2][6];
    for (int s = 0; s < 2; ++s)
        for (int a = 0; a < 6; ++a) box[s][a] = sahUnordered(nd.child_box[s][a]);
    BvhNode out;
    out.lminx = box[0][0], out.lminy = box[0][1], out.lminz = box[0][2], out.lmaxx = box[0][3], out.lmaxy = box[0][4], out.lmaxz = box[0][5];
    out.rminx = box[1][0], out.rminy = box[1][1], out.rminz = box[1][2], out.rmaxx = box[1][3], out.rmaxy = box[1][4], out.rmaxz = box[1][5];
    out.left = child[0], out.right = child[1], out.pad0 = 0, out.pad1 = 0;
    nodes[tk.id] = out;
}

// Builds the tree over nt >= 2 triangles (device arrays verts, idx) into nodes[0 .. nt-2], root = node 0.
// Synchronises the stream once per level.
inline hipError_t sahGpuBuild(const float *d_verts, const int *d_idx, int nt, float pad, int depth_limit, hipStream_t stream, BvhNode *nodes,
                              int *max_depth_out) {
    hipError_t err = hipSuccess;
    SahGpuRef *refs[2] = {nullptr, nullptr};
    SahGpuTask *tasks[2] = {nullptr, nullptr}, *small = nullptr;
    SahBigNode *big = nullptr;
    int *counters = nullptr, *chunk_node = nullptr, *chunk_left = nullptr;
    int host[6] = {1, 0, 0, 0, 0, 0};
    int n_tasks = 1, cur = 0;
    bool large_nodes = nt > kSahBigRefs;  // until a level has none: nodes only get smaller
    const int max_big = nt / kSahBigRefs + 1, max_chunks = nt / kSahChunk + max_big + 1;
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
    SAH_TRY(hipMalloc(&counters, sizeof host));
    if (large_nodes) {
        SAH_TRY(hipMalloc(&small, sizeof(SahGpuTask) * (size_t)nt));
        SAH_TRY(hipMalloc(&big, sizeof(SahBigNode) * (size_t)max_big));
        SAH_TRY(hipMalloc(&chunk_node, sizeof(int) * (size_t)max_chunks));
        SAH_TRY(hipMalloc(&chunk_left, sizeof(int) * (size_t)max_chunks));
    }
    SAH_TRY(hipMemcpyAsync(counters, host, sizeof host, hipMemcpyHostToDevice, stream));
    SAH_TRY(hipMemcpyAsync(tasks[0], &root, sizeof root, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_sah_refs, dim3((nt + 255) / 256), dim3(256), 0, stream, d_verts, d_idx, nt, pad, refs[0]);
    for (int level = 0; n_tasks > 0; ++level) {
        if (level > depth_limit + 1) {  // (the depth bound makes this unreachable)
            err = hipErrorUnknown;
            goto done;
        }
        const SahGpuTask *level_tasks = tasks[cur];
        int n_small = n_tasks, n_big = 0, n_chunks = 0;
        if (large_nodes) {
            // the level's tasks sorted into small and large ones (the counts come back: one more wait on these levels)
            SAH_TRY(hipMemsetAsync(counters + 3, 0, 3 * sizeof(int), stream));
            hipLaunchKernelGGL(k_sah_sort_tasks, dim3((n_tasks + 255) / 256), dim3(256), 0, stream, tasks[cur], n_tasks, small, big, chunk_node, counters);
            SAH_TRY(hipGetLastError());
            SAH_TRY(hipMemcpyAsync(host, counters, sizeof host, hipMemcpyDeviceToHost, stream));
            SAH_TRY(hipStreamSynchronize(stream));
            n_small = host[3], n_big = host[4], n_chunks = host[5];
            if (n_small + n_big != n_tasks || n_big > max_big || n_chunks > max_chunks) {
                err = hipErrorUnknown;
                goto done;
            }
            level_tasks = small;
            large_nodes = n_big > 0;
        }
        if (n_small > 0)
            hipLaunchKernelGGL(k_sah_level, dim3((n_small + 3) / 4), dim3(256), 0, stream, level_tasks, n_small, refs[cur], refs[cur ^ 1], tasks[cur ^ 1],
                               counters, nodes, depth_limit);
        if (n_big > 0) {
            const dim3 chunks((unsigned)n_chunks), bt(kSahBigThreads), per_node((unsigned)n_big);
            hipLaunchKernelGGL(k_sah_big_bounds, chunks, bt, 0, stream, chunk_node, big, refs[cur]);
            hipLaunchKernelGGL(k_sah_big_bins, chunks, bt, 0, stream, chunk_node, big, refs[cur], depth_limit);
            hipLaunchKernelGGL(k_sah_big_plane, per_node, dim3(64), 0, stream, big, depth_limit);
            hipLaunchKernelGGL(k_sah_big_count, chunks, bt, 0, stream, chunk_node, big, refs[cur], chunk_left);
            hipLaunchKernelGGL(k_sah_big_scan, per_node, dim3(64), 0, stream, big, chunk_left);
            hipLaunchKernelGGL(k_sah_big_scatter, chunks, bt, 0, stream, chunk_node, big, refs[cur], refs[cur ^ 1], chunk_left);
            hipLaunchKernelGGL(k_sah_big_finish, dim3((n_big + 63) / 64), dim3(64), 0, stream, big, n_big, tasks[cur ^ 1], counters, nodes);
        }
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
    void *scratch[] = {refs[0], refs[1], tasks[0], tasks[1], small, big, counters, chunk_node, chunk_left};
    for (void *p : scratch) (void)hipFree(p);
    return err;
}

}  // namespace rtx
