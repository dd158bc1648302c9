// GPU construction of the mesh acceleration structure: a linear BVH (Morton order + Karras'
// binary radix tree) over the triangles, refitted bottom-up, then emitted in the 64-byte
// two-boxes-per-node layout the traversal reads.  Replaces the reference's 7x7x7 vertex-binned
// grid (BoundingBox::init / initTriangles, src/SceneObjects.cpp:176-209), which the reference
// walks by brute force (all 343 cells, :358-361).
//
// Everything here runs once per rtx_commit(); none of it is on the per-frame path.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "rtx_device.hpp"

namespace rtx {

// Triangle boxes are padded so that the box test, done with reciprocals and FMA, can never cull a
// triangle the exact float test accepts: the rounding of (b - o) * (1/d) is a few ulp of the
// coordinates involved, far below this.
constexpr float kBvhPad = 2e-5f;

struct Aabb {
    float lo[3], hi[3];
};

// order-preserving float <-> uint mapping, so atomicMin/atomicMax on uints reduce floats
__device__ __forceinline__ unsigned floatToOrdered(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float orderedToFloat(unsigned u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// 1. gather triangles into the traversal's (v0, e1, e2) layout, per-triangle boxes, and the
//    bounds of all centroids (6 ordered uints: min xyz, max xyz)
__global__ void k_lbvh_prepare(const float *__restrict__ verts, const int *__restrict__ idx, int nt,
                               float4 *__restrict__ tris_out, Aabb *__restrict__ tri_box,
                               float *__restrict__ centroid, unsigned *__restrict__ cbounds) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float c[3] = {0, 0, 0};
    const bool live = i < nt;
    if (live) {
        const int a = idx[3 * i], b = idx[3 * i + 1], d = idx[3 * i + 2];
        float v0[3], v1[3], v2[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v0[k] = verts[3 * a + k], v1[k] = verts[3 * b + k], v2[k] = verts[3 * d + k];
        tris_out[3 * (size_t)i + 0] = make_float4(v0[0], v0[1], v0[2], 0.0f);
        tris_out[3 * (size_t)i + 1] = make_float4(v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2], 0.0f);
        tris_out[3 * (size_t)i + 2] = make_float4(v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2], 0.0f);
        Aabb bx;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bx.lo[k] = fminf(v0[k], fminf(v1[k], v2[k]));
            bx.hi[k] = fmaxf(v0[k], fmaxf(v1[k], v2[k]));
            c[k] = 0.5f * (bx.lo[k] + bx.hi[k]);
            centroid[3 * (size_t)i + k] = c[k];
        }
        tri_box[i] = bx;
    }
    // wave reduction, then one atomic per wave and bound
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float lo = live ? c[k] : kFltMax, hi = live ? c[k] : -kFltMax;
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&cbounds[k], floatToOrdered(lo));
            atomicMax(&cbounds[3 + k], floatToOrdered(hi));
        }
    }
}

__device__ __forceinline__ unsigned expandBits10(unsigned v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// 2. 30-bit Morton code of the centroid, made unique by the triangle index in the low word
__global__ void k_lbvh_morton(const float *__restrict__ centroid, const unsigned *__restrict__ cbounds, int nt,
                              unsigned long long *__restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    unsigned code = 0;
    unsigned q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lo = orderedToFloat(cbounds[k]), hi = orderedToFloat(cbounds[3 + k]);
        const float ext = hi - lo;
        float f = ext > 0.0f ? (centroid[3 * (size_t)i + k] - lo) / ext : 0.0f;
        f = fminf(fmaxf(f * 1024.0f, 0.0f), 1023.0f);
        q[k] = (unsigned)f;
    }
    code = (expandBits10(q[0]) << 2) | (expandBits10(q[1]) << 1) | expandBits10(q[2]);
    keys[i] = ((unsigned long long)code << 32) | (unsigned)i;
}

// Karras 2012: delta(i, j) = length of the common prefix of keys i and j, -1 outside the array
__device__ __forceinline__ int lbvhDelta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll(keys[i] ^ keys[j]);  // keys are unique, so the xor is never 0
}

// 3. one thread per internal node: its key range, split position, children; children learn
//    their parent.  Node numbering: internal nodes 0..n-2 (0 = root), leaf j = sorted position j.
__global__ void k_lbvh_hierarchy(const unsigned long long *__restrict__ keys, int n, int *__restrict__ left,
                                 int *__restrict__ right, int *__restrict__ parent_inner,
                                 int *__restrict__ parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (lbvhDelta(keys, n, i, i + 1) - lbvhDelta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = lbvhDelta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvhDelta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t > 0; t >>= 1)
        if (lbvhDelta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lbvhDelta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvhDelta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    // child encoding: >= 0 internal node, < 0 leaf at sorted position ~child
    const int lc = (lo == gamma) ? ~gamma : gamma;
    const int rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    left[i] = lc;
    right[i] = rc;
    if (lc < 0)
        parent_leaf[~lc] = i;
    else
        parent_inner[lc] = i;
    if (rc < 0)
        parent_leaf[~rc] = i;
    else
        parent_inner[rc] = i;
    if (i == 0) parent_inner[0] = -1;
}

// 4. bottom-up refit: one thread per leaf climbs; the second thread to reach a node merges its
//    children's boxes (the first one stops).  Boxes are handed between threads through global
//    memory, so each publish is a device-scope release (fence, then the counter atomic) and each
//    pick-up a device-scope acquire.
__global__ void k_lbvh_refit(const unsigned long long *__restrict__ keys, int n, const Aabb *__restrict__ tri_box,
                             const int *__restrict__ left, const int *__restrict__ right,
                             const int *__restrict__ parent_inner, const int *__restrict__ parent_leaf,
                             Aabb *__restrict__ node_box, unsigned *__restrict__ visits, float pad) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int node = parent_leaf[j];
    while (node >= 0) {
        __threadfence();
        const unsigned prev = atomicAdd(&visits[node], 1u);
        if (prev == 0) return;  // sibling subtree not finished yet: its thread will carry on
        __threadfence();
        Aabb bx;
#pragma unroll
        for (int k = 0; k < 3; ++k) bx.lo[k] = kFltMax, bx.hi[k] = -kFltMax;
        const int ch[2] = {left[node], right[node]};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Aabb cb;
            if (ch[c] < 0) {
                cb = tri_box[(unsigned)(keys[~ch[c]] & 0xffffffffu)];
#pragma unroll
                for (int k = 0; k < 3; ++k) cb.lo[k] -= pad, cb.hi[k] += pad;
            } else {
                const volatile Aabb *src = node_box + ch[c];
#pragma unroll
                for (int k = 0; k < 3; ++k) cb.lo[k] = src->lo[k], cb.hi[k] = src->hi[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) bx.lo[k] = fminf(bx.lo[k], cb.lo[k]), bx.hi[k] = fmaxf(bx.hi[k], cb.hi[k]);
        }
        node_box[node] = bx;
        node = parent_inner[node];
    }
}

// 5. emit traversal nodes: the two children's boxes stored in the parent, leaves resolved from
//    sorted position to triangle index.  Also measures each leaf's depth (for the stack bound).
__global__ void k_lbvh_emit(const unsigned long long *__restrict__ keys, int n, const Aabb *__restrict__ tri_box,
                            const int *__restrict__ left, const int *__restrict__ right,
                            const Aabb *__restrict__ node_box, BvhNode *__restrict__ out, float pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    Aabb b[2];
    int ch[2] = {left[i], right[i]};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (ch[c] < 0) {
            const int tri = (int)(unsigned)(keys[~ch[c]] & 0xffffffffu);
            b[c] = tri_box[tri];
#pragma unroll
            for (int k = 0; k < 3; ++k) b[c].lo[k] -= pad, b[c].hi[k] += pad;
            ch[c] = ~tri;
        } else {
            b[c] = node_box[ch[c]];
        }
    }
    BvhNode nd;
    nd.lminx = b[0].lo[0], nd.lminy = b[0].lo[1], nd.lminz = b[0].lo[2], nd.lmaxx = b[0].hi[0];
    nd.lmaxy = b[0].hi[1], nd.lmaxz = b[0].hi[2], nd.rminx = b[1].lo[0], nd.rminy = b[1].lo[1];
    nd.rminz = b[1].lo[2], nd.rmaxx = b[1].hi[0], nd.rmaxy = b[1].hi[1], nd.rmaxz = b[1].hi[2];
    nd.left = ch[0], nd.right = ch[1], nd.pad0 = 0, nd.pad1 = 0;
    out[i] = nd;
}

__global__ void k_lbvh_depth(int n, const int *__restrict__ parent_inner, const int *__restrict__ parent_leaf,
                             int *__restrict__ max_depth) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int d = 0;
    for (int node = parent_leaf[j]; node >= 0; node = parent_inner[node]) ++d;
    atomicMax(max_depth, d);
}

// single-triangle mesh: one node whose left child is the triangle and whose right box is empty
__global__ void k_lbvh_single(const Aabb *__restrict__ tri_box, BvhNode *__restrict__ out, float pad) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Aabb b = tri_box[0];
    BvhNode nd;
    nd.lminx = b.lo[0] - pad, nd.lminy = b.lo[1] - pad, nd.lminz = b.lo[2] - pad, nd.lmaxx = b.hi[0] + pad;
    nd.lmaxy = b.hi[1] + pad, nd.lmaxz = b.hi[2] + pad;
    nd.rminx = nd.rminy = nd.rminz = kFltMax;
    nd.rmaxx = nd.rmaxy = nd.rmaxz = -kFltMax;
    nd.left = ~0, nd.right = ~0, nd.pad0 = 0, nd.pad1 = 0;
    out[0] = nd;
}

}  // namespace rtx
#include "ploc_build.hpp"
#include "sah_gpu_build.hpp"
namespace rtx {

struct LbvhResult {
    float4 *tris = nullptr;   // 3 x float4 per triangle
    BvhNode *nodes = nullptr; // max(n - 1, 1) nodes
    int n_nodes = 0;
    int max_depth = 0;
};

#define RTX_HIP_TRY(expr)                      \
    do {                                       \
        hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) {                \
            err = e_;                          \
            goto done;                         \
        }                                      \
    } while (0)

// Builds the tree on `stream`.  verts/idx are device pointers.  On success the caller owns
// out->tris and out->nodes (hipFree).  Returns hipSuccess or the first failing HIP call's code.
// What to build besides the triangle array: nothing (the caller brings a tree of its own), the hierarchy of the
// Morton codes' common prefixes (Karras), agglomerative clustering along the Morton order (ploc_build.hpp), or
// the top-down binned surface-area heuristic (sah_gpu_build.hpp; bounded to `depth_limit` levels).
enum class GpuTree { kNone, kMorton, kCluster, kSah };
constexpr int kTreeGaveUp = 1 << 30;  // LbvhResult.max_depth of a GPU build that gave up: the caller builds the tree on the host
inline hipError_t lbvhBuild(const float *d_verts, const int *d_idx, int nt, GpuTree tree, int depth_limit, hipStream_t stream, LbvhResult *out) {
    const bool cluster = tree == GpuTree::kCluster;
    hipError_t err = hipSuccess;
    const int B = 256;
    const int gridT = (nt + B - 1) / B;
    Aabb *tri_box = nullptr, *node_box = nullptr;
    float *centroid = nullptr;
    unsigned *cbounds = nullptr, *visits = nullptr;
    unsigned long long *keys = nullptr, *keys_sorted = nullptr;
    int *left = nullptr, *right = nullptr, *parent_inner = nullptr, *parent_leaf = nullptr, *d_depth = nullptr;
    void *sort_tmp = nullptr;
    size_t sort_bytes = 0;
    float4 *tris = nullptr;
    BvhNode *nodes = nullptr;
    const int n_inner = nt > 1 ? nt - 1 : 1;
    const float pad = kBvhPad;
    unsigned init_bounds[6];
    {
        const float big = kFltMax;
        unsigned lo, hi;
        // ordered encodings of +max and -max
        unsigned u;
        __builtin_memcpy(&u, &big, 4);
        lo = u | 0x80000000u;
        float nbig = -big;
        __builtin_memcpy(&u, &nbig, 4);
        hi = ~u;
        for (int k = 0; k < 3; ++k) init_bounds[k] = lo, init_bounds[3 + k] = hi;
    }
    RTX_HIP_TRY(hipMalloc(&tris, sizeof(float4) * 3 * (size_t)nt));
    RTX_HIP_TRY(hipMalloc(&nodes, sizeof(BvhNode) * (size_t)n_inner));
    RTX_HIP_TRY(hipMalloc(&tri_box, sizeof(Aabb) * (size_t)nt));
    RTX_HIP_TRY(hipMalloc(&node_box, sizeof(Aabb) * (size_t)n_inner));
    RTX_HIP_TRY(hipMalloc(&centroid, sizeof(float) * 3 * (size_t)nt));
    RTX_HIP_TRY(hipMalloc(&cbounds, sizeof(unsigned) * 6));
    RTX_HIP_TRY(hipMalloc(&visits, sizeof(unsigned) * (size_t)n_inner));
    RTX_HIP_TRY(hipMalloc(&keys, sizeof(unsigned long long) * (size_t)nt));
    RTX_HIP_TRY(hipMalloc(&keys_sorted, sizeof(unsigned long long) * (size_t)nt));
    RTX_HIP_TRY(hipMalloc(&left, sizeof(int) * (size_t)n_inner));
    RTX_HIP_TRY(hipMalloc(&right, sizeof(int) * (size_t)n_inner));
    RTX_HIP_TRY(hipMalloc(&parent_inner, sizeof(int) * (size_t)n_inner));
    RTX_HIP_TRY(hipMalloc(&parent_leaf, sizeof(int) * (size_t)nt));
    RTX_HIP_TRY(hipMalloc(&d_depth, sizeof(int)));
    RTX_HIP_TRY(hipMemcpyAsync(cbounds, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, stream));
    RTX_HIP_TRY(hipMemsetAsync(visits, 0, sizeof(unsigned) * (size_t)n_inner, stream));
    RTX_HIP_TRY(hipMemsetAsync(d_depth, 0, sizeof(int), stream));
    hipLaunchKernelGGL(k_lbvh_prepare, dim3(gridT), dim3(B), 0, stream, d_verts, d_idx, nt, tris, tri_box, centroid, cbounds);
    RTX_HIP_TRY(hipGetLastError());
    if (tree == GpuTree::kNone) {
        out->max_depth = 0;
    } else if (nt == 1) {
        hipLaunchKernelGGL(k_lbvh_single, dim3(1), dim3(64), 0, stream, tri_box, nodes, pad);
        RTX_HIP_TRY(hipGetLastError());
        out->max_depth = 1;
    } else if (tree == GpuTree::kSah) {
        // (a build that gives up -- inconsistent counts from boxes that are not numbers -- is reported as a tree of
        // unbounded depth: rtx_commit then builds on the host, as it does for a tree too deep for the stack)
        const hipError_t be = sahGpuBuild(d_verts, d_idx, nt, pad, depth_limit, stream, nodes, &out->max_depth);
        if (be == hipErrorUnknown || be == hipErrorNotSupported) out->max_depth = kTreeGaveUp, (void)hipGetLastError();
        else RTX_HIP_TRY(be);
    } else {
        hipLaunchKernelGGL(k_lbvh_morton, dim3(gridT), dim3(B), 0, stream, centroid, cbounds, nt, keys);
        RTX_HIP_TRY(hipGetLastError());
        RTX_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys, keys_sorted, nt, 0, 62, stream));
        RTX_HIP_TRY(hipMalloc(&sort_tmp, sort_bytes ? sort_bytes : 16));
        RTX_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, keys, keys_sorted, nt, 0, 62, stream));
        if (cluster) {
            const hipError_t be = plocBuild<Aabb>(keys_sorted, tri_box, nt, pad, stream, nodes, &out->max_depth);
            if (be == hipErrorUnknown || be == hipErrorNotSupported) out->max_depth = kTreeGaveUp, (void)hipGetLastError();
            else RTX_HIP_TRY(be);
            goto built;
        }
        hipLaunchKernelGGL(k_lbvh_hierarchy, dim3(gridT), dim3(B), 0, stream, keys_sorted, nt, left, right, parent_inner, parent_leaf);
        RTX_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_lbvh_refit, dim3(gridT), dim3(B), 0, stream, keys_sorted, nt, tri_box, left, right, parent_inner, parent_leaf, node_box, visits, pad);
        RTX_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_lbvh_emit, dim3(gridT), dim3(B), 0, stream, keys_sorted, nt, tri_box, left, right, node_box, nodes, pad);
        RTX_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_lbvh_depth, dim3(gridT), dim3(B), 0, stream, nt, parent_inner, parent_leaf, d_depth);
        RTX_HIP_TRY(hipGetLastError());
        RTX_HIP_TRY(hipMemcpyAsync(&out->max_depth, d_depth, sizeof(int), hipMemcpyDeviceToHost, stream));
    }
built:
    RTX_HIP_TRY(hipStreamSynchronize(stream));
    out->tris = tris, out->nodes = nodes, out->n_nodes = n_inner;
    tris = nullptr, nodes = nullptr;
done:
    void *scratch[] = {tris, nodes, tri_box, node_box, centroid, cbounds, visits, keys, keys_sorted, left, right,
                       parent_inner, parent_leaf, d_depth, sort_tmp};
    for (void *p : scratch) (void)hipFree(p);
    return err;
}

}  // namespace rtx
