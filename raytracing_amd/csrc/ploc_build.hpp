// GPU construction of the mesh tree by parallel locally-ordered clustering (PLOC, Meister & Bittner 2018) on top of
// the Morton order that lbvh_build.hpp produces: the clusters -- at first the triangles in Morton order -- each
// look for the neighbour within `kPlocRadius` places whose union with them has the smallest surface area; pairs
// that chose each other merge into a new inner node that takes the place of the first of the two; the list is
// compacted; and so on until one cluster is left.  Bottom-up agglomeration by surface area gives trees of the
// quality of a top-down surface-area-heuristic build, which the plain Morton-order tree does not reach, at a
// cost of a few dozen small launches per rtx_commit.
//
// Like the other builders this only decides how the walk is steered; what a ray hits does not depend on it.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include "rtx_device.hpp"

namespace rtx {

constexpr int kPlocRadius = 32;  // (8 .. 100 measured on the bunny: node visits of a frame vary by 4 %, 32 is the best)

__device__ __forceinline__ float plocUnionArea(const float *a_lo, const float *a_hi, const float *b_lo, const float *b_hi) {
    const float dx = fmaxf(a_hi[0], b_hi[0]) - fminf(a_lo[0], b_lo[0]);
    const float dy = fmaxf(a_hi[1], b_hi[1]) - fminf(a_lo[1], b_lo[1]);
    const float dz = fmaxf(a_hi[2], b_hi[2]) - fminf(a_lo[2], b_lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

// boxes of the leaves (cluster j = the triangle at sorted position j), padded like every box of the tree
template <typename Box>
__global__ void k_ploc_leaves(const unsigned long long *__restrict__ keys, int n, const Box *__restrict__ tri_box, float pad,
                              Box *__restrict__ box, int *__restrict__ cluster) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Box b = tri_box[(unsigned)(keys[j] & 0xffffffffu)];
#pragma unroll
    for (int k = 0; k < 3; ++k) b.lo[k] -= pad, b.hi[k] += pad;
    box[j] = b;
    cluster[j] = j;
}

// 1. every cluster's nearest neighbour (by surface area of the union) among the kPlocRadius clusters on either side
template <typename Box>
__global__ void k_ploc_nearest(const int *__restrict__ cluster, int m, const Box *__restrict__ box, int *__restrict__ nearest, int radius) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const Box me = box[cluster[i]];
    float best = kFltMax;
    int best_j = -1;
    const int lo = i - radius > 0 ? i - radius : 0, hi = i + radius < m - 1 ? i + radius : m - 1;
    for (int j = lo; j <= hi; ++j) {
        if (j == i) continue;
        const Box other = box[cluster[j]];
        const float a = plocUnionArea(me.lo, me.hi, other.lo, other.hi);
        if (a < best) best = a, best_j = j;  // (the lower index wins ties: both sides of a tie must agree on it)
    }
    nearest[i] = best_j;
}

// 2. pairs that chose each other merge: the new node takes the lower one's place, the upper one's place is dropped
template <typename Box>
__global__ void k_ploc_merge(const int *__restrict__ cluster, int m, const int *__restrict__ nearest, int n_leaves, Box *__restrict__ box,
                             int *__restrict__ left, int *__restrict__ right, int *__restrict__ parent, int *__restrict__ n_inner,
                             int *__restrict__ merged, unsigned char *__restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int j = nearest[i];
    int id = cluster[i];
    unsigned char k = 1;
    if (j >= 0 && nearest[j] == i) {
        if (i < j) {
            const int a = id, b = cluster[j];
            const int node = n_leaves + atomicAdd(n_inner, 1);
            const Box ba = box[a], bb = box[b];
            Box u;
#pragma unroll
            for (int c = 0; c < 3; ++c) u.lo[c] = fminf(ba.lo[c], bb.lo[c]), u.hi[c] = fmaxf(ba.hi[c], bb.hi[c]);
            box[node] = u;
            left[node - n_leaves] = a, right[node - n_leaves] = b;
            parent[a] = node, parent[b] = node;
            id = node;
        } else {
            k = 0;
        }
    }
    merged[i] = id;
    keep[i] = k;
}

// 3. the traversal's nodes.  Inner node `id` (n_leaves + creation number) goes to index n_inner - 1 - creation
// number: the root, created last, to index 0, every parent before its children.
template <typename Box>
__global__ void k_ploc_emit(const unsigned long long *__restrict__ keys, int n_leaves, int n_inner, const Box *__restrict__ box,
                            const int *__restrict__ left, const int *__restrict__ right, BvhNode *__restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_inner) return;
    int ch[2] = {left[k], right[k]};
    Box b[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        b[c] = box[ch[c]];
        ch[c] = ch[c] < n_leaves ? ~(int)(unsigned)(keys[ch[c]] & 0xffffffffu) : n_inner - 1 - (ch[c] - n_leaves);
    }
    BvhNode nd;
    nd.lminx = b[0].lo[0], nd.lminy = b[0].lo[1], nd.lminz = b[0].lo[2], nd.lmaxx = b[0].hi[0];
    nd.lmaxy = b[0].hi[1], nd.lmaxz = b[0].hi[2], nd.rminx = b[1].lo[0], nd.rminy = b[1].lo[1];
    nd.rminz = b[1].lo[2], nd.rmaxx = b[1].hi[0], nd.rmaxy = b[1].hi[1], nd.rmaxz = b[1].hi[2];
    nd.left = ch[0], nd.right = ch[1], nd.pad0 = 0, nd.pad1 = 0;
    out[n_inner - 1 - k] = nd;
}

// depth of the deepest leaf (number of inner nodes above it)
__global__ void k_ploc_depth(int n_leaves, const int *__restrict__ parent, int root, int *__restrict__ max_depth) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    int d = 0;
    for (int node = j; node != root; node = parent[node]) ++d;
    atomicMax(max_depth, d);
}

// Builds the tree over the n >= 2 triangles whose sorted Morton keys (triangle index in the low word) are `keys`.
// `nodes` has room for n - 1 entries.  Synchronises the stream (the cluster count comes back once per round).
// hipErrorNotSupported: the clusters would not merge (see below); nothing usable is in `nodes` then.
template <typename Box>
inline hipError_t plocBuild(const unsigned long long *keys, const Box *tri_box, int n, float pad, hipStream_t stream, BvhNode *nodes,
                            int *max_depth_out) {
    hipError_t err = hipSuccess;
    const int B = 256;
    Box *box = nullptr;
    int *cluster[2] = {nullptr, nullptr}, *nearest = nullptr, *merged = nullptr, *left = nullptr, *right = nullptr, *parent = nullptr;
    int *counters = nullptr;  // [0] inner nodes made, [1] clusters kept by the last compaction, [2] deepest leaf
    unsigned char *keep = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    int host[3] = {0, 0, 0};
    int m = n, cur = 0;
#define PLOC_TRY(expr)            \
    do {                          \
        err = (expr);             \
        if (err != hipSuccess) goto done; \
    } while (0)
    PLOC_TRY(hipMalloc(&box, sizeof(Box) * (size_t)(2 * n - 1)));
    PLOC_TRY(hipMalloc(&cluster[0], sizeof(int) * (size_t)n));
    PLOC_TRY(hipMalloc(&cluster[1], sizeof(int) * (size_t)n));
    PLOC_TRY(hipMalloc(&nearest, sizeof(int) * (size_t)n));
    PLOC_TRY(hipMalloc(&merged, sizeof(int) * (size_t)n));
    PLOC_TRY(hipMalloc(&left, sizeof(int) * (size_t)(n - 1)));
    PLOC_TRY(hipMalloc(&right, sizeof(int) * (size_t)(n - 1)));
    PLOC_TRY(hipMalloc(&parent, sizeof(int) * (size_t)(2 * n - 1)));
    PLOC_TRY(hipMalloc(&counters, sizeof(int) * 3));
    PLOC_TRY(hipMalloc(&keep, (size_t)n));
    PLOC_TRY(hipMemsetAsync(counters, 0, sizeof(int) * 3, stream));
    PLOC_TRY(hipcub::DeviceSelect::Flagged(nullptr, tmp_bytes, merged, keep, cluster[1], counters + 1, n, stream));
    PLOC_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    hipLaunchKernelGGL(k_ploc_leaves<Box>, dim3((n + B - 1) / B), dim3(B), 0, stream, keys, n, tri_box, pad, box, cluster[0]);
    // Every round costs a synchronisation and a pass over the clusters; a healthy mesh loses a third or more of its
    // clusters per round.  Many equal union areas (coincident triangles) or boxes that are not numbers make a round
    // merge next to nothing: such a build is given up (hipErrorNotSupported) and the caller builds the tree another way.
    for (int round = 0; m > 1; ++round) {
        if (round >= 96) {
            err = hipErrorNotSupported;
            goto done;
        }
        const dim3 grid((m + B - 1) / B);
        hipLaunchKernelGGL(k_ploc_nearest<Box>, grid, dim3(B), 0, stream, cluster[cur], m, box, nearest, kPlocRadius);
        hipLaunchKernelGGL(k_ploc_merge<Box>, grid, dim3(B), 0, stream, cluster[cur], m, nearest, n, box, left, right, parent, counters, merged, keep);
        PLOC_TRY(hipGetLastError());
        PLOC_TRY(hipcub::DeviceSelect::Flagged(tmp, tmp_bytes, merged, keep, cluster[cur ^ 1], counters + 1, m, stream));
        PLOC_TRY(hipMemcpyAsync(host, counters, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
        PLOC_TRY(hipStreamSynchronize(stream));
        if (host[1] >= m || host[1] < 1 || (long long)(m - host[1]) * 64 < m) {  // (a round merges at least the globally closest pair)
            err = hipErrorNotSupported;
            goto done;
        }
        m = host[1], cur ^= 1;
    }
    if (host[0] != n - 1) {
        err = hipErrorUnknown;
        goto done;
    }
    hipLaunchKernelGGL(k_ploc_emit<Box>, dim3((n - 1 + B - 1) / B), dim3(B), 0, stream, keys, n, n - 1, box, left, right, nodes);
    hipLaunchKernelGGL(k_ploc_depth, dim3((n + B - 1) / B), dim3(B), 0, stream, n, parent, 2 * n - 2, counters + 2);
    PLOC_TRY(hipGetLastError());
    PLOC_TRY(hipMemcpyAsync(host, counters, sizeof(int) * 3, hipMemcpyDeviceToHost, stream));
    PLOC_TRY(hipStreamSynchronize(stream));
    *max_depth_out = host[2];
done:
#undef PLOC_TRY
    void *scratch[] = {box, cluster[0], cluster[1], nearest, merged, left, right, parent, counters, keep, tmp};
    for (void *p : scratch) (void)hipFree(p);
    return err;
}

}  // namespace rtx
