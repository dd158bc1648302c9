// Host-side construction of the mesh BVH by the surface-area heuristic (binned, top-down), for
// meshes small enough that a few milliseconds on the host at rtx_commit() buy a better tree than
// the Morton-order build in lbvh_build.hpp: rays visit fewer nodes, and -- what matters more for a
// frame that is only a few milliseconds long -- the longest walk of a wavefront gets shorter.
// Output is the same 64-byte two-boxes-per-node array (root = node 0, children >= 0 inner,
// < 0 triangle ~child), in depth-first order, so the traversal does not know which builder ran.
//
// Replaces, like lbvh_build.hpp, the reference's 7x7x7 vertex-binned grid (BoundingBox::init /
// initTriangles, src/SceneObjects.cpp:176-209).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <vector>

#include "rtx_device.hpp"

namespace rtx {

struct SahBox {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) lo[k] = FLT_MAX, hi[k] = -FLT_MAX;
    }
    void grow(const SahBox &o) {
        for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], o.lo[k]), hi[k] = std::max(hi[k], o.hi[k]);
    }
    void grow(const float p[3]) {
        for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], p[k]), hi[k] = std::max(hi[k], p[k]);
    }
    float halfArea() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx < 0 ? 0.0f : dx * dy + dy * dz + dz * dx;
    }
};

struct SahRef {
    SahBox box;  // already padded
    float c[3];
    int tri;
};

class SahBuilder {
  public:
    // verts: xyz per vertex, idx: three vertex numbers per triangle, pad: widening of every triangle box
    SahBuilder(const std::vector<float> &verts, const std::vector<int> &idx, float pad) {
        const size_t nt = idx.size() / 3;
        refs_.resize(nt);
        for (size_t t = 0; t < nt; ++t) {
            SahRef &r = refs_[t];
            r.box.reset();
            for (int v = 0; v < 3; ++v) r.box.grow(&verts[3 * (size_t)idx[3 * t + v]]);
            for (int k = 0; k < 3; ++k) {
                r.c[k] = 0.5f * (r.box.lo[k] + r.box.hi[k]);
                r.box.lo[k] -= pad, r.box.hi[k] += pad;
            }
            r.tri = (int)t;
        }
    }

    // the same over ready-made boxes (leaf ~child = index of the box); centroid = middle of the box
    explicit SahBuilder(const std::vector<SahBox> &boxes) {
        refs_.resize(boxes.size());
        for (size_t t = 0; t < boxes.size(); ++t) {
            SahRef &r = refs_[t];
            r.box = boxes[t];
            for (int k = 0; k < 3; ++k) r.c[k] = 0.5f * (r.box.lo[k] + r.box.hi[k]);
            r.tri = (int)t;
        }
    }

    // Fills nodes (max(nt - 1, 1) of them) and returns the depth of the deepest leaf.
    int build(std::vector<BvhNode> &nodes, int depth_limit) {
        const size_t nt = refs_.size();
        nodes.assign(nt > 1 ? nt - 1 : 1, BvhNode{});
        next_ = 0;
        max_depth_ = 0;
        depth_limit_ = depth_limit;
        if (nt == 1) {  // one node whose left child is the triangle and whose right box is empty
            BvhNode &nd = nodes[0];
            setBox(nd, 0, refs_[0].box);
            SahBox empty;
            empty.reset();
            setBox(nd, 1, empty);
            nd.left = ~0, nd.right = ~0;
            return 1;
        }
        SahBox all;
        split(nodes, 0, nt, 1, all);
        return max_depth_;
    }

  private:
    static constexpr int kBins = 32;

    static void setBox(BvhNode &nd, int side, const SahBox &b) {
        if (side == 0)
            nd.lminx = b.lo[0], nd.lminy = b.lo[1], nd.lminz = b.lo[2], nd.lmaxx = b.hi[0], nd.lmaxy = b.hi[1], nd.lmaxz = b.hi[2];
        else
            nd.rminx = b.lo[0], nd.rminy = b.lo[1], nd.rminz = b.lo[2], nd.rmaxx = b.hi[0], nd.rmaxy = b.hi[1], nd.rmaxz = b.hi[2];
    }

    // Builds the subtree over refs_[b, e) (e - b >= 2) as inner node `next_`; returns its id and box.
    int split(std::vector<BvhNode> &nodes, size_t b, size_t e, int depth, SahBox &box_out) {
        const int id = next_++;
        SahBox cb;
        cb.reset();
        for (size_t i = b; i < e; ++i) cb.grow(refs_[i].c);
        size_t mid = b;
        const size_t n = e - b;
        // remaining levels must be able to hold the range even when every later split is a median split
        int levels_needed = 0;
        while (((size_t)1 << levels_needed) < n) ++levels_needed;
        const bool forced_median = depth + levels_needed >= depth_limit_;
        if (!forced_median && n > 2) {
            float best = FLT_MAX;
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                const float ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0.0f)) continue;
                SahBox bins[kBins];
                int count[kBins] = {};
                for (auto &x : bins) x.reset();
                const float scale = kBins / ext;
                for (size_t i = b; i < e; ++i) {
                    const int k = std::min(kBins - 1, std::max(0, (int)((refs_[i].c[axis] - cb.lo[axis]) * scale)));
                    bins[k].grow(refs_[i].box);
                    ++count[k];
                }
                float right_area[kBins];
                SahBox acc;
                acc.reset();
                for (int k = kBins - 1; k > 0; --k) {
                    acc.grow(bins[k]);
                    right_area[k] = acc.halfArea();
                }
                acc.reset();
                int left_n = 0;
                for (int k = 0; k + 1 < kBins; ++k) {
                    acc.grow(bins[k]);
                    left_n += count[k];
                    if (left_n == 0 || left_n == (int)n) continue;
                    const float cost = acc.halfArea() * left_n + right_area[k + 1] * (float)(n - left_n);
                    if (cost < best) best = cost, best_axis = axis, best_bin = k;
                }
            }
            if (best_axis >= 0) {
                const float ext = cb.hi[best_axis] - cb.lo[best_axis], scale = kBins / ext, lo = cb.lo[best_axis];
                auto it = std::partition(refs_.begin() + b, refs_.begin() + e, [&](const SahRef &r) {
                    return std::min(kBins - 1, std::max(0, (int)((r.c[best_axis] - lo) * scale))) <= best_bin;
                });
                mid = (size_t)(it - refs_.begin());
            }
        }
        if (mid == b || mid == e) {  // no usable plane (coincident centroids, two triangles, depth limit): halve
            int axis = 0;
            for (int k = 1; k < 3; ++k)
                if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
            mid = b + n / 2;
            std::nth_element(refs_.begin() + b, refs_.begin() + mid, refs_.begin() + e,
                             [axis](const SahRef &x, const SahRef &y) { return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.tri < y.tri); });
        }
        SahBox boxes[2];
        int child[2];
        const size_t lo_[2] = {b, mid}, hi_[2] = {mid, e};
        for (int s = 0; s < 2; ++s) {
            if (hi_[s] - lo_[s] == 1) {
                boxes[s] = refs_[lo_[s]].box;
                child[s] = ~refs_[lo_[s]].tri;
                max_depth_ = std::max(max_depth_, depth);
            } else {
                child[s] = split(nodes, lo_[s], hi_[s], depth + 1, boxes[s]);
            }
        }
        BvhNode &nd = nodes[id];
        setBox(nd, 0, boxes[0]);
        setBox(nd, 1, boxes[1]);
        nd.left = child[0], nd.right = child[1], nd.pad0 = 0, nd.pad1 = 0;
        box_out = boxes[0];
        box_out.grow(boxes[1]);
        return id;
    }

    std::vector<SahRef> refs_;
    int next_ = 0, max_depth_ = 0, depth_limit_ = 32;
};

}  // namespace rtx
