// Four children per node.  The walks (rtx_trace.hpp) are chains of dependent node fetches: a ray cannot ask for a
// node before it has tested the one above it, and what a wave waits for is the latency of those fetches, not the
// arithmetic.  A node with four children halves the length of the chain.  The binary trees the builders make
// (sah_gpu_build.hpp, lbvh_build.hpp, ploc_build.hpp, sah_build.hpp) are collapsed here, on the GPU, level by level
// from the root: a wide node starts with the two children of its binary node and then twice replaces the inner
// child with the largest surface area by that child's own two children, so that up to three binary nodes become
// one wide node and the slots go to the boxes a ray is most likely to meet.
//
// Layout (128 bytes, one cache line): the children's boxes plane by plane -- lo.x of all four, lo.y, lo.z, hi.x,
// hi.y, hi.z, 16 bytes each -- so that a lane loads, per axis, the four near planes and the four far planes for
// the sign of its own ray with an offset of its own (no minimum / maximum per box and axis); then the four child
// ids (>= 0 inner node, < 0 leaf ~id) and the number of children.  An unused slot has an empty box, which no ray
// passes, and repeats the first child's id, so that even arithmetic gone wrong (NaN from origins beyond 1e8)
// could only make a walk repeat itself.
//
// The traversal stack a tree needs is worked out exactly, bottom-up: a node with m children pushes up to m - 1 of
// them before it descends into the first, so need(node) = max over its children, sorted by falling need, of
// need(child i) + m - 1 - i, with need(leaf) = 0.
//
// Replaces, like the builders, the reference's 7x7x7 vertex-binned grid and its brute-force walk
// (src/SceneObjects.cpp:176-209, 350-380).  The tree only steers the walk.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "rtx_device.hpp"

namespace rtx {

struct Bvh4Task {
    int src, dst;  // the binary node to collapse, the wide node to write
};

__device__ __forceinline__ float bvh4HalfArea(const float b[6]) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx < 0.0f ? -1.0f : dx * dy + dy * dz + dz * dx;
}

// counters: [0] wide nodes allocated, [1] tasks queued for the next level
__global__ __launch_bounds__(256) void k_bvh4_level(const BvhNode *__restrict__ bvh2, const Bvh4Task *__restrict__ tasks, int n_tasks,
                                                    Bvh4Task *__restrict__ next, int *__restrict__ counters, Bvh4Node *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    const Bvh4Task tk = tasks[t];
    float box[4][6];
    int id[4];
    int n = 2;
    auto take = [&](const BvhNode &nd, int left_slot, int right_slot) {
        box[left_slot][0] = nd.lminx, box[left_slot][1] = nd.lminy, box[left_slot][2] = nd.lminz;
        box[left_slot][3] = nd.lmaxx, box[left_slot][4] = nd.lmaxy, box[left_slot][5] = nd.lmaxz;
        box[right_slot][0] = nd.rminx, box[right_slot][1] = nd.rminy, box[right_slot][2] = nd.rminz;
        box[right_slot][3] = nd.rmaxx, box[right_slot][4] = nd.rmaxy, box[right_slot][5] = nd.rmaxz;
        id[left_slot] = nd.left, id[right_slot] = nd.right;
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        id[k] = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) box[k][a] = kFltMax, box[k][3 + a] = -kFltMax;
    }
    take(bvh2[tk.src], 0, 1);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        int widest = -1;
        float area = -1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = bvh4HalfArea(box[k]);
            if (k < n && id[k] >= 0 && a > area) area = a, widest = k;
        }
        if (widest < 0) break;
        const BvhNode nd = bvh2[id[widest]];
        // (static indices keep the arrays in registers)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k == widest) {
#pragma unroll
                for (int j = 2; j < 4; ++j)
                    if (j == n) take(nd, k, j);
            }
        ++n;
    }
    int inner = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) inner += k < n && id[k] >= 0 ? 1 : 0;
    int node_at = 0, task_at = 0;
    if (inner) node_at = atomicAdd(&counters[0], inner), task_at = atomicAdd(&counters[1], inner);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < n && id[k] >= 0) {
            next[task_at++] = Bvh4Task{id[k], node_at};
            id[k] = node_at++;
        }
        if (k >= n) id[k] = id[0];
    }
    float4 *o = reinterpret_cast<float4 *>(out + tk.dst);
#pragma unroll
    for (int a = 0; a < 6; ++a) o[a] = make_float4(box[0][a], box[1][a], box[2][a], box[3][a]);
    reinterpret_cast<int4 *>(o)[6] = make_int4(id[0], id[1], id[2], id[3]);
    reinterpret_cast<int4 *>(o)[7] = make_int4(n, 0, 0, 0);
}

// wide nodes [first, last) are one level of the tree; the levels below are done.  need[i] = stack entries the
// walk of node i's subtree can have in use at once; height[i] = levels below and including it.
__global__ __launch_bounds__(256) void k_bvh4_need(const Bvh4Node *__restrict__ nodes, int first, int last, unsigned char *__restrict__ need,
                                                   unsigned char *__restrict__ height) {
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= last) return;
    const int4 ch = nodes[i].child;
    const int id[4] = {ch.x, ch.y, ch.z, ch.w};
    const int m = nodes[i].pad.x;
    int v[4], h = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = -1;
        if (k >= m) continue;
        v[k] = id[k] >= 0 ? need[id[k]] : 0;
        if (id[k] >= 0) h = max(h, (int)height[id[k]]);
    }
    // falling order (unused slots, -1, last)
#define RTX_BVH4_SWAP(a, b)            \
    if (v[a] < v[b]) {                 \
        const int t_ = v[a];           \
        v[a] = v[b], v[b] = t_;        \
    }
    RTX_BVH4_SWAP(0, 1) RTX_BVH4_SWAP(2, 3) RTX_BVH4_SWAP(0, 2) RTX_BVH4_SWAP(1, 3) RTX_BVH4_SWAP(1, 2)
#undef RTX_BVH4_SWAP
    int worst = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < m) worst = max(worst, v[k] + m - 1 - k);
    need[i] = (unsigned char)min(worst, 255);
    height[i] = (unsigned char)min(h + 1, 255);
}

struct Bvh4Result {
    Bvh4Node *nodes = nullptr;  // the caller's (hipFree)
    int n_nodes = 0;
    int stack_need = 0;  // stack entries a walk can have in use at once
    int height = 0;      // levels of wide nodes
};

// Collapses the binary tree d_bvh2 (root = node 0, n_bvh2 nodes) into a tree of wide nodes.  Synchronises the
// stream once per level of the result.
inline hipError_t bvh4Collapse(const BvhNode *d_bvh2, int n_bvh2, hipStream_t stream, Bvh4Result *out) {
    hipError_t err = hipSuccess;
    Bvh4Node *wide = nullptr, *exact = nullptr;
    Bvh4Task *tasks[2] = {nullptr, nullptr};
    int *counters = nullptr;
    unsigned char *need = nullptr, *height = nullptr;
    int host[2] = {1, 0};
    unsigned char top[2] = {0, 0};
    const Bvh4Task root{0, 0};
    std::vector<int> level_first{0};
    int n_tasks = 1, cur = 0;
#define BVH4_TRY(expr)                     \
    do {                                   \
        err = (expr);                      \
        if (err != hipSuccess) goto done;  \
    } while (0)
    BVH4_TRY(hipMalloc(&wide, sizeof(Bvh4Node) * (size_t)n_bvh2));
    BVH4_TRY(hipMalloc(&tasks[0], sizeof(Bvh4Task) * (size_t)n_bvh2));
    BVH4_TRY(hipMalloc(&tasks[1], sizeof(Bvh4Task) * (size_t)n_bvh2));
    BVH4_TRY(hipMalloc(&counters, sizeof host));
    BVH4_TRY(hipMemcpyAsync(counters, host, sizeof host, hipMemcpyHostToDevice, stream));
    BVH4_TRY(hipMemcpyAsync(tasks[0], &root, sizeof root, hipMemcpyHostToDevice, stream));
    while (n_tasks > 0) {
        if ((int)level_first.size() > n_bvh2 + 1) {  // (every level consumes a binary node: unreachable)
            err = hipErrorUnknown;
            goto done;
        }
        hipLaunchKernelGGL(k_bvh4_level, dim3((n_tasks + 255) / 256), dim3(256), 0, stream, d_bvh2, tasks[cur], n_tasks, tasks[cur ^ 1], counters,
                           wide);
        BVH4_TRY(hipGetLastError());
        BVH4_TRY(hipMemcpyAsync(host, counters, sizeof host, hipMemcpyDeviceToHost, stream));
        BVH4_TRY(hipStreamSynchronize(stream));
        level_first.push_back(host[0] - host[1]);  // where the level just queued begins
        n_tasks = host[1];
        BVH4_TRY(hipMemsetAsync(counters + 1, 0, sizeof(int), stream));
        cur ^= 1;
    }
    // level_first = {0, 1, first of level 2, ..., n_nodes}: the last launch queued nothing
    if (host[0] < 1 || host[0] > n_bvh2) {
        err = hipErrorUnknown;
        goto done;
    }
    BVH4_TRY(hipMalloc(&need, (size_t)host[0]));
    BVH4_TRY(hipMalloc(&height, (size_t)host[0]));
    for (int l = (int)level_first.size() - 2; l >= 0; --l) {
        const int first = level_first[l], last = level_first[l + 1];
        if (last <= first) continue;
        hipLaunchKernelGGL(k_bvh4_need, dim3((last - first + 255) / 256), dim3(256), 0, stream, wide, first, last, need, height);
    }
    BVH4_TRY(hipGetLastError());
    BVH4_TRY(hipMemcpyAsync(&top[0], need, 1, hipMemcpyDeviceToHost, stream));
    BVH4_TRY(hipMemcpyAsync(&top[1], height, 1, hipMemcpyDeviceToHost, stream));
    // the nodes move to an allocation of their own size
    BVH4_TRY(hipMalloc(&exact, sizeof(Bvh4Node) * (size_t)host[0]));
    BVH4_TRY(hipMemcpyAsync(exact, wide, sizeof(Bvh4Node) * (size_t)host[0], hipMemcpyDeviceToDevice, stream));
    BVH4_TRY(hipStreamSynchronize(stream));
    out->nodes = exact, out->n_nodes = host[0], out->stack_need = top[0], out->height = top[1];
    exact = nullptr;
done:
#undef BVH4_TRY
    void *scratch[] = {wide, exact, tasks[0], tasks[1], counters, need, height};
    for (void *p : scratch) (void)hipFree(p);
    return err;
}

}  // namespace rtx
