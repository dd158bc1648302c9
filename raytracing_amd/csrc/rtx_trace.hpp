// The trace loop proper: closest hit over all primitive categories, shadow queries, Phong
// lighting, the bounded per-lane ray stack that replaces traceRay's recursion, sky lookup.
#pragma once
#include "rtx_device.hpp"
#include "rtx_refmath.hpp"

namespace rtx {

// Per-lane statistics; they exist only in the STATS instantiation (static indices keep them in
// registers) and are summed per wave into the global Counters at the end of the kernel.
template <bool STATS>
struct LaneCounters {
    unsigned v[CNT_SLOTS] = {};
};
template <>
struct LaneCounters<false> {};

// true in exactly one of the lanes that execute the call together
RTX_DEV bool firstActiveLane() {
    const unsigned long long m = __ballot(1);
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(0xffffffffu, __builtin_amdgcn_mbcnt_lo(0xffffffffu, 0u));
    return lane == (unsigned)__builtin_ctzll(m);
}

#define RTX_COUNT(slot) \
    if constexpr (STATS) lc.v[slot]++
#define RTX_COUNT_N(slot, n) \
    if constexpr (STATS) lc.v[slot] += (n)

// ---------------------------------------------------------------------------------------------
// Mesh: LBVH traversal.  Returns the closest front-facing hit under Triangle::rayIntersect with
// t < best_t (strict, as modelsIntersect's `curDist < modelsDist && curDist < minDist`,
// src/SceneObjects.cpp:365-366).  `stack` points at this lane's column of an LDS array whose rows
// are `stride` ints apart, so the 64 lanes of a wave hit 64 different banks.
// The walk stops at the first hit nearer than t_exit (shadow rays; closest-hit rays pass a
// negative t_exit) and then returns true.
// ---------------------------------------------------------------------------------------------
// The gate of the mesh: Model::boxIntersect on the padded vertex bounds, src/SceneObjects.cpp:356 --
// the cheap filter first, the reference's own slab test (six IEEE divisions) only for candidates.
template <bool STATS>
RTX_DEV bool meshGate(const DevScene &s, V3 orig, V3 dir, V3 inv, LaneCounters<STATS> &lc) {
    if (!slabMaybe(s.mesh_lo, s.mesh_hi, orig, inv)) return false;
    RTX_COUNT(CNT_BOX_TESTS);
    float gate_near = 0.0f;
    if (!boxSlab(s.mesh_lo, s.mesh_hi, orig, dir, gate_near)) return false;
    RTX_COUNT(CNT_BOX_HITS);
    return true;
}

constexpr int kNoCell = 0x7fffffff, kKeyOpen = 0x7ffffffe;

// Smallest key among the triangle's cells whose box the ray passes under the reference's own slab
// test; the cell boxes are formed as BoundingBox::init forms them, src/SceneObjects.cpp:183-192.
RTX_DEV int cellKey(const DevScene &s, int4 keys, V3 orig, V3 dir, V3 inv) {
    int key = kNoCell;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const int k = c == 0 ? keys.x : (c == 1 ? keys.y : keys.z);
        if (k < 0 || k >= key) continue;
        const int cell = k >> 22;
        const int ci = cell / 49, cj = (cell / 7) % 7, ck = cell % 7;
        const float lo[3] = {s.mesh_lo[0] + s.grid_step[0] * (float)ci, s.mesh_lo[1] + s.grid_step[1] * (float)cj,
                             s.mesh_lo[2] + s.grid_step[2] * (float)ck};
        const float hi[3] = {s.mesh_lo[0] + s.grid_step[0] * (float)(ci + 1), s.mesh_lo[1] + s.grid_step[1] * (float)(cj + 1),
                             s.mesh_lo[2] + s.grid_step[2] * (float)(ck + 1)};
        float cell_near = 0.0f;
        if (slabMaybe(lo, hi, orig, inv) && boxSlab(lo, hi, orig, dir, cell_near)) key = k;
    }
    return key;
}

// One step of a walk over a tree of two-box nodes: tests the children of inner node `cur` against the ray (plane
// distances t_lo = lo * ia + a, t_hi = hi * ib + b per axis; nothing beyond `bound` counts), goes to the nearer
// child that is hit and pushes the other; kWalkDone when neither is hit and the stack is empty.  Box tests only
// steer a walk; they use reciprocals and FMA freely.
constexpr int kWalkDone = (int)0x80000000;
RTX_DEV int walkPop(const int *stack, int stride, int &sp) {
    int next = kWalkDone;
    if (sp > 0) {
        --sp;
        next = stack[sp * stride];
    }
    return next;
}
template <bool WIDEN>
RTX_DEV int walkStep(const BvhNode *nodes, int cur, float iax, float ibx, float iay, float iby, float iaz, float ibz, float ax, float bx,
                     float ay, float by, float az, float bz, float bound, int *stack, int stride, int &sp) {
    // (the node's address as the array's, which is the same in all lanes, plus 32 bits of offset: no address arithmetic
    // in vector registers; trees stay below 4 GB)
    const float4 *np = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(nodes) + ((unsigned)cur << 6));
    static_assert(sizeof(BvhNode) == 64, "node offsets are index << 6");
    const float4 a = np[0], b = np[1], c = np[2];
    const int2 ch = *reinterpret_cast<const int2 *>(np + 3);
    // t_far is widened by 2 ulp where the boxes themselves are all the margin there is (WIDEN)
    constexpr float widen = WIDEN ? 1.0000004f : 1.0f;
    float tnL, tfL, tnR, tfR;
    {
        float t0 = __builtin_fmaf(a.x, iax, ax), t1 = __builtin_fmaf(a.w, ibx, bx);
        float t2 = __builtin_fmaf(a.y, iay, ay), t3 = __builtin_fmaf(b.x, iby, by);
        float t4 = __builtin_fmaf(a.z, iaz, az), t5 = __builtin_fmaf(b.y, ibz, bz);
        tnL = fmaxf(fmaxf(fminf(t0, t1), fminf(t2, t3)), fmaxf(fminf(t4, t5), 0.0f));
        tfL = fminf(fminf(fmaxf(t0, t1), fmaxf(t2, t3)), fminf(fmaxf(t4, t5), bound)) * widen;
    }
    {
        float t0 = __builtin_fmaf(b.z, iax, ax), t1 = __builtin_fmaf(c.y, ibx, bx);
        float t2 = __builtin_fmaf(b.w, iay, ay), t3 = __builtin_fmaf(c.z, iby, by);
        float t4 = __builtin_fmaf(c.x, iaz, az), t5 = __builtin_fmaf(c.w, ibz, bz);
        tnR = fmaxf(fmaxf(fminf(t0, t1), fminf(t2, t3)), fmaxf(fminf(t4, t5), 0.0f));
        tfR = fminf(fminf(fmaxf(t0, t1), fmaxf(t2, t3)), fminf(fmaxf(t4, t5), bound)) * widen;
    }
    const bool hitL = tnL <= tfL, hitR = tnR <= tfR;
    // near child (or the only one hit) next, the far one onto the stack; written without nested branches: the
    // store is harmless when nothing is pushed (the slot is free)
    const bool both = hitL && hitR;
    const bool leftNext = hitL && (!hitR || tnL <= tnR);
    stack[sp * stride] = leftNext ? ch.y : ch.x;
    sp += both ? 1 : 0;
    int next = leftNext ? ch.x : ch.y;
    if (!(hitL || hitR)) next = walkPop(stack, stride, sp);
    return next;
}

// The inner loop of a walk that keeps its leaves: node steps; a leaf that is met goes into `pend` if that is free and
// the walk goes on.  It ends, for all lanes of the wave together, when a lane that holds a leaf meets the next one or
// no lane has an inner node left: the caller then tests the kept leaves.  (A loop of its own so that what the leaf
// tests change -- the nearest hit, the bound -- is not carried around it: a dozen register moves a trip.)
template <bool STATS, bool WIDEN>
RTX_DEV void walkToLeaves(const BvhNode *nodes, float iax, float ibx, float iay, float iby, float iaz, float ibz, float ax, float bx, float ay,
                          float by, float az, float bz, float bound, int &cur, int &sp, int &pend, int *stack, int stride,
                          LaneCounters<STATS> &lc) {
    for (;;) {
        if (cur >= 0) {
            RTX_COUNT(CNT_NODE_TESTS);
            if constexpr (STATS) lc.v[CNT_NODE_TRIPS] += firstActiveLane() ? 1u : 0u;
            cur = walkStep<WIDEN>(nodes, cur, iax, ibx, iay, iby, iaz, ibz, ax, bx, ay, by, az, bz, bound, stack, stride, sp);
        }
        if (cur < 0 && cur != kWalkDone && pend == kWalkDone) {
            pend = cur;
            cur = walkPop(stack, stride, sp);
        }
        if (__ballot(cur < 0 && cur != kWalkDone) != 0ull || __ballot(cur >= 0) == 0ull) return;
    }
}

// The triangle of leaf `leaf` of the mesh's tree against the ray: Triangle::rayIntersect and the bookkeeping of the
// reference's cell walk.  True: the walk is over (a hit nearer than t_exit: a shadow ray's occluder).
template <bool STATS>
RTX_DEV bool meshLeafTest(const DevScene &s, V3 orig, V3 dir, V3 inv, int leaf, float &best_t, int &best_tri, int &best_key, float t_exit,
                          LaneCounters<STATS> &lc) {
    const int ti = ~leaf;
    const float4 *tp = s.mesh_tris + 3 * (size_t)ti;
    const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
    // fetched with the triangle, not after the test: a dependent load would double the latency
    int4 keys = make_int4(-1, -1, -1, -1);
    if (s.mesh_cells) keys = s.mesh_cells[ti];
    RTX_COUNT(CNT_TRI_TESTS);
    if constexpr (STATS) lc.v[CNT_TRI_TRIPS] += firstActiveLane() ? 1u : 0u;
    float t = 0.0f;
    bool past_u = false;
    const bool th = triHit(v3(p0.x, p0.y, p0.z), v3(p1.x, p1.y, p1.z), v3(p2.x, p2.y, p2.z), orig, dir, t,
                           STATS ? &past_u : nullptr);
    if constexpr (STATS) lc.v[CNT_TRI_FULL] += past_u ? 1u : 0u;
    if (th && t <= best_t) {
        // The reference tests a triangle only for rays that pass (Cube::rayIntersect) a grid
        // cell holding one of its vertices, and of two hits at the same distance it keeps the
        // one its i,j,k cell walk reaches first.  key = that order (cell, position in the
        // cell's list); kNoCell = the ray passes none of the triangle's cells.
        int key = ti;
        if (s.mesh_cells) {
            key = kNoCell;
            // Shortcut for the common case: the hit point lies well inside one of the
            // triangle's cells, so the ray certainly passes that cell's box (the margin is
            // orders of magnitude above the rounding of the slab test; directions with a
            // component below the slab test's parallel threshold, far hits and exact ties
            // take the long way).  The order key is then left open (kKeyOpen) and only
            // worked out if a tie ever needs it.
            if (t < best_t && t < s.grid_fast_tmax && fabsf(dir.x) >= kEps && fabsf(dir.y) >= kEps && fabsf(dir.z) >= kEps) {
                const V3 p = orig + dir * t;
                const float rx = (p.x - s.mesh_lo[0]) * s.grid_inv_step[0], ry = (p.y - s.mesh_lo[1]) * s.grid_inv_step[1],
                            rz = (p.z - s.mesh_lo[2]) * s.grid_inv_step[2];
                const float fx = floorf(rx), fy = floorf(ry), fz = floorf(rz);
                const float lo_m = fminf(fminf(rx - fx, ry - fy), rz - fz), hi_m = fmaxf(fmaxf(rx - fx, ry - fy), rz - fz);
                if (lo_m >= 1e-3f && hi_m <= 0.999f && fminf(fminf(fx, fy), fz) >= 0.0f && fmaxf(fmaxf(fx, fy), fz) <= 6.0f) {
                    const int cell = ((int)fx * 7 + (int)fy) * 7 + (int)fz;
                    if ((keys.x >= 0 && (keys.x >> 22) == cell) || (keys.y >= 0 && (keys.y >> 22) == cell) ||
                        (keys.z >= 0 && (keys.z >> 22) == cell))
                        key = kKeyOpen;
                }
            }
            if (key == kNoCell) key = cellKey(s, keys, orig, dir, inv);
        }
        bool take = key != kNoCell && t < best_t;
        if (key != kNoCell && t == best_t && best_tri >= 0) {  // a tie: both keys exactly
            if (best_key == kKeyOpen) best_key = cellKey(s, s.mesh_cells[best_tri], orig, dir, inv);
            take = key < best_key;
        }
        if (take) {
            best_key = key;
            best_t = t;
            best_tri = ti;
            if (t < t_exit) return true;
        }
    }
    return false;
}

// DEFER picks the form of the walk.  false: two alternating phases, each a loop the whole wave runs together --
// descend through inner nodes until this lane holds a leaf (lanes that found theirs idle until the rest have), then
// one triangle test for every lane that holds one.  true: one loop in which a lane that comes upon a leaf keeps it
// for later and walks on; the kept triangles are tested, by all lanes that hold one, when a lane holds one and
// comes upon the next, or when no lane has an inner node left.  The winner is the smallest (distance, key) pair
// whatever the order of the tests, and a test made later only lets the walk look at boxes that an earlier hit would
// have culled.  The second form makes fewer, fuller node trips and more triangle trips: it is the faster one in the
// level-by-level kernels (a wave of walkers only) and the slower one in k_render (measured: +4-7 % frame time).
// `cur` >= 0 is an inner node, < 0 is leaf ~cur, kWalkDone means the stack ran dry.  A single-triangle mesh has an
// inner root with one leaf.
template <bool STATS, bool DEFER>
RTX_DEV bool meshIntersect(const DevScene &s, V3 orig, V3 dir, V3 inv, float &best_t, int &best_tri, float t_exit,
                           int *stack, int stride, LaneCounters<STATS> &lc) {
    // Node boxes were padded when the tree was built, and t_far is widened by 2 ulp, so a triangle the exact test
    // would accept is never culled by rounding here.
    const float ix = inv.x, iy = inv.y, iz = inv.z;  // clamped reciprocals: no inf - inf below
    const float ox = -orig.x * ix, oy = -orig.y * iy, oz = -orig.z * iz;
    int sp = 0;
    int cur = 0;
    int best_key = kNoCell;  // order in which the reference's walk reaches best_tri
    const auto step = [&]() {
        RTX_COUNT(CNT_NODE_TESTS);
        if constexpr (STATS) lc.v[CNT_NODE_TRIPS] += firstActiveLane() ? 1u : 0u;
        cur = walkStep<true>(s.bvh, cur, ix, ix, iy, iy, iz, iz, ox, ox, oy, oy, oz, oz, best_t, stack, stride, sp);
    };
    if constexpr (DEFER) {
        int pend = kWalkDone;
        for (;;) {
            walkToLeaves<STATS, true>(s.bvh, ix, ix, iy, iy, iz, iz, ox, ox, oy, oy, oz, oz, best_t, cur, sp, pend, stack, stride, lc);
            if (pend != kWalkDone) {
                if (meshLeafTest<STATS>(s, orig, dir, inv, pend, best_t, best_tri, best_key, t_exit, lc)) return true;
                pend = kWalkDone;
            }
            if (cur == kWalkDone) break;
        }
    } else {
        for (;;) {
            while (cur >= 0) step();
            // the lanes that hold a leaf test it, and go on testing while what they pop is a leaf again
            while (cur != kWalkDone && cur < 0) {
                if (meshLeafTest<STATS>(s, orig, dir, inv, cur, best_t, best_tri, best_key, t_exit, lc)) return true;
                cur = walkPop(stack, stride, sp);
            }
            if (cur == kWalkDone) break;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// Spheres beyond the LDS staging limit: spheresIntersect (src/SceneObjects.cpp:289-302) over a BVH2
// of the spheres' boxes instead of the whole list.  The test at a leaf is the reference's own
// (sphereHit); the winner is the nearest sphere and, among equal distances, the one that comes
// first in the list, which is what the reference's loop with its strict `curDist < spheresDist`
// keeps.
//
// What the walk may cull.  Sphere::rayIntersect compares dSquared = dot(L, L) - t0 * t0 with r * r,
// where L = centre - origin; the rounding of that difference is proportional to |L|^2 (below
// 18 * 2^-24 * |L|^2 for the operation order of sphereHit: 5 from dot(L, L), 9 from t0 * t0, 4 from
// the length of the direction), so a ray can be accepted although it
// passes the centre at up to r + kSphereSlack * t0, t0 being the distance of its closest approach.
// For a far, small sphere that is well outside the sphere's box.  The boxes are therefore widened
// per ray, by delta = kSphereSlack * (largest t0 that can still matter + largest radius): no hit can
// matter beyond the far side of the box around all spheres, nor (once a hit is known) beyond that
// hit's distance plus a radius.
// ---------------------------------------------------------------------------------------------
#ifndef RTX_SPHERE_SLACK  // (a build with a smaller value must fail the grazing-ray tests)
#define RTX_SPHERE_SLACK 1.2e-3f
#endif
constexpr float kSphereSlack = RTX_SPHERE_SLACK;  // > sqrt(18 * 2^-24) = 1.04e-3

// Largest closest-approach distance that can matter for a ray from `orig` (no hit known yet): the far
// side of the box around all spheres, and for a shadow ray (t_exit > 0, just short of the light's
// distance or the far clip) what lies beyond t_exit -- a hit out there leaves the light visible,
// exactly as no hit does.
RTX_DEV float sphereReach(const DevScene &s, V3 orig, float t_exit) {
    const V3 d = orig - v3(s.sphere_root_c[0], s.sphere_root_c[1], s.sphere_root_c[2]);
    const float reach = 1.01f * (sqrtf(dot(d, d)) + s.sphere_root_h);
    return t_exit > 0.0f ? fminf(reach, 1.02f * (t_exit + s.sphere_rmax)) : reach;
}

// One ray's walk over the sphere tree, in a form that can stop and go on (sphereWalkSome).
struct SphereWalker {
    float iax, ax, ibx, bx, iay, ay, iby, by, iaz, az, ibz, bz;  // plane distances: t_lo = lo * ia + a, t_hi = hi * ib + b
    float bound, reach;
    int cur, sp, pend;  // >= 0 inner node, < 0 sphere ~cur, kWalkDone: nothing left; stack fill; a sphere whose test is put off
};
RTX_DEV bool sphereWalkerLive(const SphereWalker &w) { return w.cur != kWalkDone || w.pend != kWalkDone; }

RTX_DEV SphereWalker sphereWalkerStart(const DevScene &s, V3 orig, V3 dir, V3 inv, float reach, float minDist);

// Walks on, all lanes of the wave together, until fewer than `min_live` lanes have anything left to visit; a lane
// that is through (a shadow ray's occluder found, or nothing left) idles meanwhile.  The same steps as sphereWalk.
template <bool STATS>
RTX_DEV void sphereWalkSome(const DevScene &s, SphereWalker &w, V3 orig, V3 dir, float t_exit, float &minDist, int &index, int *stack,
                            int stride, unsigned min_live, LaneCounters<STATS> &lc) {
    for (;;) {
        if ((unsigned)__popcll(__ballot(sphereWalkerLive(w))) < min_live) return;
        walkToLeaves<STATS, false>(s.sphere_bvh, w.iax, w.ibx, w.iay, w.iby, w.iaz, w.ibz, w.ax, w.bx, w.ay, w.by, w.az, w.bz, w.bound, w.cur, w.sp, w.pend,
                            stack, stride, lc);
        if (w.pend != kWalkDone) {
            const int i = ~w.pend;
            w.pend = kWalkDone;
            const float4 sph = s.spheres[i];
            RTX_COUNT(CNT_SPHERE_TESTS);
            float curDist = 0.0f;
            if (sphereHit(v3(sph.x, sph.y, sph.z), sph.w, orig, dir, curDist) && (curDist < minDist || (curDist == minDist && i < index))) {
                minDist = curDist;
                index = i;
                if (curDist < t_exit) w.cur = kWalkDone, w.sp = 0;
                w.bound = fminf(w.reach, 1.01f * (minDist + s.sphere_rmax));
            }
        }
    }
}

RTX_DEV SphereWalker sphereWalkerStart(const DevScene &s, V3 orig, V3 dir, V3 inv, float reach, float minDist) {
    SphereWalker w;
    w.reach = reach;
    w.bound = fminf(reach, 1.01f * (minDist + s.sphere_rmax));  // no candidate's closest approach lies beyond
    // Plane distances of the widened boxes, per axis t_lo = lo * ia + a and t_hi = hi * ib + b.  The widening a
    // candidate needs grows with the distance of its closest approach, so the boxes are widened by
    // kSphereSlack * (t + rmax) at ray parameter t instead of by the largest such value everywhere: the ray point
    // o + t d lies inside the widened slab of an axis when  t (d + k) >= lo - k rmax - o  and  t (d - k) <= hi + k rmax - o,
    // two half lines in t like those of the plain slab test, with the reciprocals of d + k and d - k in place of
    // that of d.  (Where both bounds are negative -- the box lies behind the origin on this axis -- their order may
    // be the other way round; the interval then ends below zero either way.)  An axis the ray runs almost parallel
    // to (|d| <= 2 k: d - k and d + k differ in sign or nearly vanish) gets the uniform widening by the largest value,
    // (lo - delta - o) * inv = lo * inv + a and (hi + delta - o) * inv = hi * inv + b.  Measured on 10 000 spheres:
    // 4 % fewer node tests and 23 % fewer sphere tests than uniform widening on all axes.
    const float k = kSphereSlack, kr = kSphereSlack * s.sphere_rmax, delta = kSphereSlack * (w.bound + s.sphere_rmax);
    const auto axis = [&](float o, float d, float iv, float &ia, float &a, float &ib, float &b) {
        if (fabsf(d) > 2.0f * k) {
            ia = __builtin_amdgcn_rcpf(d + k), ib = __builtin_amdgcn_rcpf(d - k);
            a = -(o + kr) * ia, b = -(o - kr) * ib;
        } else {
            ia = ib = iv;
            a = -(o + delta) * iv, b = -(o - delta) * iv;
        }
    };
    axis(orig.x, dir.x, inv.x, w.iax, w.ax, w.ibx, w.bx);
    axis(orig.y, dir.y, inv.y, w.iay, w.ay, w.iby, w.by);
    axis(orig.z, dir.z, inv.z, w.iaz, w.az, w.ibz, w.bz);
    w.sp = 0, w.cur = 0, w.pend = kWalkDone;
    return w;
}

template <bool STATS>
RTX_DEV void sphereWalk(const DevScene &s, V3 orig, V3 dir, V3 inv, float reach, float t_exit, float &minDist, int &index,
                        int *stack, int stride, LaneCounters<STATS> &lc) {
    SphereWalker w = sphereWalkerStart(s, orig, dir, inv, reach, minDist);
    for (;;) {
        // A lane that comes upon a leaf keeps it for later and walks on.  The kept spheres are tested, by all lanes
        // that hold one, when a lane holds one and comes upon the next, or when no lane has an inner node left: the
        // sphere test then runs for some twenty lanes instead of eight per trip, and the lanes that used to wait for
        // it walk (10 000 spheres: 116.6 -> 104.2 ms a frame).  The result is the nearest hit, the lowest index among
        // equals, whatever the order of the tests; a test made later only lets the walk look at boxes that an earlier
        // hit would have culled.
        walkToLeaves<STATS, false>(s.sphere_bvh, w.iax, w.ibx, w.iay, w.iby, w.iaz, w.ibz, w.ax, w.bx, w.ay, w.by, w.az, w.bz, w.bound, w.cur, w.sp, w.pend,
                            stack, stride, lc);
        if (w.pend != kWalkDone) {
            const int i = ~w.pend;
            w.pend = kWalkDone;
            const float4 sph = s.spheres[i];
            RTX_COUNT(CNT_SPHERE_TESTS);
            float curDist = 0.0f;
            if (sphereHit(v3(sph.x, sph.y, sph.z), sph.w, orig, dir, curDist) && (curDist < minDist || (curDist == minDist && i < index))) {
                minDist = curDist;
                index = i;
                if (curDist < t_exit) return;
                w.bound = fminf(w.reach, 1.01f * (minDist + s.sphere_rmax));
            }
        }
        if (!sphereWalkerLive(w)) return;
    }
}

// ---------------------------------------------------------------------------------------------
// Scene::sceneIntersect, src/Scene.cpp:29-58: categories in the reference's order, each accepting
// only a strictly nearer hit.  It comes in three parts so that a wave can run the middle one -- the
// mesh walk, by far the longest and the one few rays need -- for many lanes at once:
//   castBegin: spheres, loose triangles, boxes, the mesh gate; says whether the walk is needed
//   castMesh : the walk
//   castEnd  : water plane, islands, then hit point / normal / material of the winner
// Shadow rays (shadow = true) only need the distance; their categories are skipped as soon as any
// hit nearer than t_exit is known (the caller chooses t_exit so that the reference's own test
// `distance(shadowPt, shadowOrig) < distance(light, point)` is then certain).  Closest-hit rays
// pass shadow = false and a negative t_exit.  `shadow` is a per-lane value: the lanes of a wave
// cast both kinds side by side.
// ---------------------------------------------------------------------------------------------
struct CastState {
    float minDist;
    int kind;   // what was hit last: 0 none, 1 sphere, 2 loose triangle, 3 box, 4 mesh triangle, 5 water, 6 island
    int index;
};

// The gate in front of the walk over the sphere tree: false when the ray misses the box around everything that has
// one.  object_bounds holds the spheres' geometric boxes: the gate needs the same per-ray slack as the walk.
RTX_DEV bool sphereTreeGate(const DevScene &s, const SmallScene &q, V3 orig, V3 inv, float t_exit, float &reach) {
    reach = sphereReach(s, orig, t_exit);
    const float delta = kSphereSlack * (reach + s.sphere_rmax);
    const float lo[3] = {q.object_bounds[0] - delta, q.object_bounds[1] - delta, q.object_bounds[2] - delta};
    const float hi[3] = {q.object_bounds[3] + delta, q.object_bounds[4] + delta, q.object_bounds[5] + delta};
    return slabMaybe(lo, hi, orig, inv);
}

template <bool STATS, bool WIDE>
RTX_DEV bool castLoose(const DevScene &s, const SmallScene &q, V3 orig, V3 dir, V3 inv, float t_exit, float minDist, int kind, int index,
                       CastState &cs, LaneCounters<STATS> &lc);

// WIDE = the general build of the kernels: sphere sets beyond the LDS staging limit (behind their
// BVH), several samples per pixel.  The default build leaves both out, the code and the registers.
template <bool STATS, bool WIDE>
RTX_DEV bool castBegin(const DevScene &s, const SmallScene &q, V3 orig, V3 dir, float t_exit, CastState &cs, int *stack, int stride,
                       LaneCounters<STATS> &lc) {
    float minDist = kFltMax;
    int kind = 0, index = -1;
    const V3 inv = safeInverse(dir);
    cs.minDist = minDist, cs.kind = kind, cs.index = index;
    bool walked = false;
    if constexpr (WIDE) {
        if (s.sphere_bvh) {
            float reach;
            if (!sphereTreeGate(s, q, orig, inv, t_exit, reach)) return false;
            sphereWalk<STATS>(s, orig, dir, inv, reach, t_exit, minDist, index, stack, stride, lc);
            if (index >= 0) kind = 1;
            walked = true;
        }
    }
    // One filter for everything that has a bounding box: most rays of an open scene miss it all.  (For every
    // launch the host sizes object_bounds and sphere_bounds with the slack that the rounding of
    // Sphere::rayIntersect calls for at the largest distance a ray of that launch can start from -- see
    // sphereWalk and sphereGateBounds -- so the two filters never reject a ray whose sphere test would pass.)
    if (!walked && !slabMaybe(q.object_bounds, q.object_bounds + 3, orig, inv)) return false;

    // spheresIntersect, src/SceneObjects.cpp:289-302
    if (!walked && s.n_spheres > 0 && slabMaybe(q.sphere_bounds, q.sphere_bounds + 3, orig, inv)) {
        RTX_COUNT_N(CNT_SPHERE_TESTS, s.n_spheres);
        for (int i = 0; i < s.n_spheres; ++i) {
            const float4 sp = q.spheres[i];
            float curDist = 0.0f;
            if (sphereHit(v3(sp.x, sp.y, sp.z), sp.w, orig, dir, curDist) && curDist < minDist) {
                minDist = curDist;
                kind = 1, index = i;
            }
        }
    }
    return castLoose<STATS, WIDE>(s, q, orig, dir, inv, t_exit, minDist, kind, index, cs, lc);
}

// The part of castBegin after the spheres: loose triangles, boxes, the mesh gate (minDist, kind, index: the nearest
// sphere, if any).
template <bool STATS, bool WIDE>
RTX_DEV bool castLoose(const DevScene &s, const SmallScene &q, V3 orig, V3 dir, V3 inv, float t_exit, float minDist, int kind, int index,
                       CastState &cs, LaneCounters<STATS> &lc) {
    // trianglesIntersect, :304-317
    if (s.n_tris > 0 && !(minDist < t_exit) && slabMaybe(q.tri_bounds, q.tri_bounds + 3, orig, inv)) {
        RTX_COUNT_N(CNT_LOOSE_TRI_TESTS, s.n_tris);
        auto test = [&](const float *t, int i) {
            const V3 v0 = v3(t[0], t[1], t[2]), v1 = v3(t[3], t[4], t[5]), v2 = v3(t[6], t[7], t[8]);
            float curDist = 0.0f;
            if (triHit(v0, v1 - v0, v2 - v0, orig, dir, curDist) && curDist < minDist) {
                minDist = curDist;
                kind = 2, index = i;
            }
        };
        const int n_staged = s.n_tris < kMaxSmallPrims ? s.n_tris : kMaxSmallPrims;
        for (int i = 0; i < n_staged; ++i) test(q.tris + 9 * i, i);
        if constexpr (WIDE) {
#pragma unroll 1
            for (int i = kMaxSmallPrims; i < s.n_tris; ++i) test(s.tris + 9 * (size_t)i, i);
        }
    }
    // cubesIntersect, :382-397
    if (s.n_boxes > 0 && !(minDist < t_exit) && slabMaybe(q.box_bounds, q.box_bounds + 3, orig, inv)) {
        auto test = [&](const float *b, int i) {
            const float lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[3], b[4], b[5]};
            if (!slabMaybe(lo, hi, orig, inv)) return;
            RTX_COUNT(CNT_BOX_TESTS);
            float curDist = 0.0f;
            if (boxSlab(lo, hi, orig, dir, curDist) && curDist < minDist) {
                minDist = curDist;
                kind = 3, index = i;
            }
        };
        const int n_staged = s.n_boxes < kMaxSmallPrims ? s.n_boxes : kMaxSmallPrims;
        for (int i = 0; i < n_staged; ++i) test(q.boxes + 6 * i, i);
        if constexpr (WIDE) {
#pragma unroll 1
            for (int i = kMaxSmallPrims; i < s.n_boxes; ++i) test(s.boxes + 6 * (size_t)i, i);
        }
    }
    cs.minDist = minDist, cs.kind = kind, cs.index = index;
    // modelsIntersect, :350-380
    return s.n_mesh_tris > 0 && !(minDist < t_exit) && meshGate<STATS>(s, orig, dir, inv, lc);
}

template <bool STATS, bool DEFER = false>
RTX_DEV void castMesh(const DevScene &s, V3 orig, V3 dir, float t_exit, CastState &cs, int *stack, int stride, LaneCounters<STATS> &lc) {
    int tri = -1;
    float t = cs.minDist;
    meshIntersect<STATS, DEFER>(s, orig, dir, safeInverse(dir), t, tri, t_exit, stack, stride, lc);
    if (tri >= 0) cs.minDist = t, cs.kind = 4, cs.index = tri;
}

// Returns `minDist < far_clip`; minDist itself comes back in min_dist.
template <bool STATS, bool WIDE>
RTX_DEV bool castEnd(const DevScene &s, const SmallScene &q, V3 orig, V3 dir, bool shadow, float t_exit, const CastState &cs,
                     Hit &h, float &min_dist, LaneCounters<STATS> &lc) {
#pragma clang fp contract(off)  // hit points and texel coordinates: geometry, never contracted (see rtx_device.hpp)
    float minDist = cs.minDist;
    int kind = cs.kind, index = cs.index;
    // water plane, src/Scene.cpp:38-54
    if (s.water_w > 0 && !(minDist < t_exit)) {
        RTX_COUNT(CNT_PLANE_TESTS);
        float waterDist = kFltMax;
        if (planeHit(s.water_y, orig, dir, waterDist) && waterDist < minDist) {
            minDist = waterDist;
            kind = 5;
        }
    }
    // islandsIntersect, src/SceneObjects.cpp:399-419 with Island::rayIntersect :273-287.  Every
    // island that passes overrides the previous one (the loop compares with minDist, which it
    // does not update); minDist takes the last one's distance afterwards.
    if (!(minDist < t_exit)) {
        float islandsDist = kFltMax;
        float dist = 0.0f;
        RTX_COUNT_N(CNT_PLANE_TESTS, s.n_islands > 0 ? 1 : 0);
        if (s.n_islands > 0 && planeHit(s.island_y, orig, dir, dist)) {
            RTX_COUNT_N(CNT_ISLAND_TESTS, s.n_islands);
            const V3 pt = orig + dir * dist;
            // the rectangle around all islands first (padded; the exact conditions follow per island)
            const bool near_islands = pt.x > q.island_rect[0] && pt.x < q.island_rect[1] && pt.z > q.island_rect[2] && pt.z < q.island_rect[3];
            for (int i = 0; near_islands && i < s.n_islands; ++i) {
                const float4 is = islandAt<WIDE>(s, q, i);  // cx, cz, a, b
                // the bounding-rectangle conditions come first in the reference's && chain as well; the
                // two divisions of the ellipse test are only paid inside the rectangle
                if (pt.x > is.x - is.z && pt.x < is.x + is.z && pt.z > is.y - is.w && pt.z < is.y + is.w && dist < minDist) {
                    const float xa = (pt.x - is.x) / is.z, zb = (pt.z - is.y) / is.w;
                    if (xa * xa + zb * zb <= 1.0f) {
                        islandsDist = dist;
                        kind = 6, index = i;
                    }
                }
            }
        }
        minDist = fminf(islandsDist, minDist);
    }
    min_dist = minDist;
    if (shadow) return minDist < s.far_clip;

    // ---- resolve the winner: hit point, normal, material ----
    if (kind == 0) return false;
    h.point = orig + dir * minDist;
    const int mat = hitMaterialIndex<WIDE>(s, q, kind, index);
    if (kind == 1) {
        float4 sp;
        if (WIDE && s.sphere_bvh) sp = s.spheres[index];
        else sp = ldsFloat4(q.spheres + index);  // (see lightAt for the form)
        h.N = normalize(h.point - v3(sp.x, sp.y, sp.z));
    } else if (kind == 2) {
        const Floats9 t = triAt<WIDE>(s, q, index);
        const V3 v0 = v3(t[0], t[1], t[2]), v1 = v3(t[3], t[4], t[5]), v2 = v3(t[6], t[7], t[8]);
        h.N = normalize(triNormalDir(v1 - v0, v2 - v0));
    } else if (kind == 3) {
        const Floats6 b = boxAt<WIDE>(s, q, index);
        const float lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[3], b[4], b[5]};
        h.N = normalize(boxNormalDir(lo, hi, h.point));
    } else if (kind == 4) {
        const float4 *tp = s.mesh_tris + 3 * (size_t)index;
        const float4 p1 = tp[1], p2 = tp[2];
        h.N = normalize(triNormalDir(v3(p1.x, p1.y, p1.z), v3(p2.x, p2.y, p2.z)));
    } else if (kind == 5) {
        h.N = v3(0.0f, 1.0f, 0.0f);
    } else {
        h.N = v3(0.0f, 1.0f, 0.0f);
    }
    h.m = materialAt<WIDE>(s, q, mat);
    h.mat_index = mat;
    if (kind == 5) {
        // texture tiles every `scale` units, nearest texel, blend (tex*0.3 + diffuse)/2: src/Scene.cpp:46-51
        float x = h.point.x / s.water_scale, z = h.point.z / s.water_scale;
        float xx = x - floorf(x), zz = z - floorf(z);
        unsigned xpos = (unsigned)(xx * s.water_w), zpos = (unsigned)(zz * s.water_h);
        // xx rounds up to 1.0 for tiny negative x; the reference then reads one element further
        // (past the end on the last row).  Clamp to the last texel instead.
        size_t idx = (size_t)zpos * s.water_w + xpos;
        const size_t last = (size_t)s.water_w * s.water_h - 1;
        idx = idx < last ? idx : last;
        const unsigned char *p = s.water + idx * 3;
        // texel = byte / 255.0f, a true divide (src/Scene.cpp:246), unlike the sky's multiply
        const float tr = (float)p[0] / 255.0f, tg = (float)p[1] / 255.0f, tb = (float)p[2] / 255.0f;
        h.m.dr = (tr * 0.3f + h.m.dr) / 2.0f;
        h.m.dg = (tg * 0.3f + h.m.dg) / 2.0f;
        h.m.db = (tb * 0.3f + h.m.db) / 2.0f;
    } else if (kind == 6) {
        // checker and dots: src/SceneObjects.cpp:409-415
        const int parity = truncToInt(h.point.x + h.point.z) & 1;
        const float *c0 = parity ? q.light_green : q.dark_green;
        const float *c1 = parity ? q.dark_green : q.light_green;
        float x = h.point.x, fx = roundf(h.point.x);
        float z = ceilf(h.point.x + h.point.z) - fx - 0.5f;
        const float rad = 0.15f;
        const bool dot_ = (x - fx) * (x - fx) + (h.point.z - z) * (h.point.z - z) < rad * rad;
        const float *c = dot_ ? c1 : c0;
        h.m.dr = c[0], h.m.dg = c[1], h.m.db = c[2];
    }
    return minDist < s.far_clip;
}

// (DEFER: the form of the mesh walk, see meshIntersect)
template <bool STATS, bool WIDE = false, bool DEFER = false>
RTX_DEV bool sceneIntersect(const DevScene &s, const SmallScene &q, V3 orig, V3 dir, bool shadow, float t_exit, Hit &h,
                            float &min_dist, int *stack, int stride, LaneCounters<STATS> &lc) {
    CastState cs;
    if (castBegin<STATS, WIDE>(s, q, orig, dir, t_exit, cs, stack, stride, lc)) castMesh<STATS, DEFER>(s, orig, dir, t_exit, cs, stack, stride, lc);
    return castEnd<STATS, WIDE>(s, q, orig, dir, shadow, t_exit, cs, h, min_dist, lc);
}

// Sky::getColorAt, src/SceneObjects.cpp:440-445.  atan2 and acos bind to the float overloads in
// that translation unit; the two quotients are then formed in double and truncated to int; the
// row shift wraps in uint32 before the modulo.  Texel = byte * (1/255.0f) (:429-434).
RTX_DEV V3 skyColor(const DevScene &s, V3 dir) {
#pragma clang fp contract(off)
    const int width = s.sky_w, height = s.sky_h;
    const float at = refAtan2f(dir.z, dir.x);
    const float ac = refAcosf(dir.y);
    int a = truncToIntD(((double)at / (2 * 3.14159265358979323846) + .5) * width);
    a = max(0, min(width - 1, a));
    int b = truncToIntD((double)ac / 3.14159265358979323846 * height);
    b = max(0, min(height - 1, b));
    const unsigned place = (unsigned)a + ((unsigned)(b - s.sky_shift) % (unsigned)height) * (unsigned)width;
    const size_t idx = (size_t)(long)(int)place % ((size_t)width * height);
    const unsigned char *p = s.sky + idx * 3;
    const float k = 1 / 255.0f;
    return v3((float)p[0] * k, (float)p[1] * k, (float)p[2] * k);
}

// ---------------------------------------------------------------------------------------------
// Scene::traceRay + Scene::calcLightning, src/Scene.cpp:60-106, as a state machine that advances
// by ONE RAY per step: either the closest-hit ray of the next node of the call tree, or one shadow
// ray of the node being shaded.  A wave interleaves the paths of 64 pixels at that granularity, so
// a lane whose ray left the scene goes on to its next ray at once instead of idling while its
// neighbours cast their shadow rays.
//
// A frame is pushed for every hit that has at least one child to wait for; children are traced
// reflect first, refract second, and the sum is formed left to right exactly as at :103-105:
//     ((diffuse*Ld*a0 + 1*Ls*a1) + reflect*a2) + refract*a3.
// With pruning on, a child whose weight is exactly 0 is not traced and its (exactly zero)
// contribution not added; x + 0*c == x for every finite c, so the result is unchanged.
// ---------------------------------------------------------------------------------------------
struct FrameHead {
    V3 acc;  // partial sum so far
    float a2, a3;
    int state;  // bit0: waiting for reflect, bit1: refract still to do, bit2: waiting for refract, bits 8..: depth of this node
};
struct FrameRay {
    V3 orig, dir;  // pending refract ray (only nodes with both children have one)
};

// The frames of one lane.  The heads of the first `lds_slots` frames live in LDS -- field k of frame j at
// lds[(6 * j + k) * lds_stride], consecutive lanes in consecutive dwords -- the heads beyond them and the
// pending refract rays in the lane's scratch memory (dynamically indexed arrays cannot stay in registers).
// A kernel that has no LDS to spare passes lds_slots = 0.
struct FrameSpill {
    FrameHead heads[kMaxDepth];
    FrameRay rays[kMaxDepth];
};
// (LDS through a pointer that says so: through a plain pointer the compiler merges the LDS read and the scratch read of
// frameHead into one flat load of a selected address, which waits for both memory pipes)
using LdsFloat = __attribute__((address_space(3))) float;
struct FrameStack {
    LdsFloat *lds;
    int lds_slots, lds_stride;
    FrameHead *heads;
    FrameRay *rays;
};

RTX_DEV FrameStack frameStack(FrameSpill &spill, float *lds = nullptr, int lds_slots = 0, int lds_stride = 0) {
    return FrameStack{(LdsFloat *)lds, lds_slots, lds_stride, spill.heads, spill.rays};
}

RTX_DEV FrameHead frameHead(const FrameStack &fs, int j) {
    if (j < fs.lds_slots) {
        const LdsFloat *p = fs.lds + 6 * j * fs.lds_stride;
        FrameHead f;
        f.acc = v3(p[0], p[fs.lds_stride], p[2 * fs.lds_stride]);
        f.a2 = p[3 * fs.lds_stride], f.a3 = p[4 * fs.lds_stride];
        f.state = __float_as_int(p[5 * fs.lds_stride]);
        return f;
    }
    return fs.heads[j];
}

RTX_DEV void frameSetHead(FrameStack &fs, int j, const FrameHead &f) {
    if (j < fs.lds_slots) {
        LdsFloat *p = fs.lds + 6 * j * fs.lds_stride;
        p[0] = f.acc.x, p[fs.lds_stride] = f.acc.y, p[2 * fs.lds_stride] = f.acc.z;
        p[3 * fs.lds_stride] = f.a2, p[4 * fs.lds_stride] = f.a3;
        p[5 * fs.lds_stride] = __int_as_float(f.state);
    } else {
        fs.heads[j] = f;
    }
}

// after the reflect child: only the partial sum and the state have changed
RTX_DEV void frameSetSum(FrameStack &fs, int j, const FrameHead &f) {
    if (j < fs.lds_slots) {
        LdsFloat *p = fs.lds + 6 * j * fs.lds_stride;
        p[0] = f.acc.x, p[fs.lds_stride] = f.acc.y, p[2 * fs.lds_stride] = f.acc.z;
        p[5 * fs.lds_stride] = __int_as_float(f.state);
    } else {
        fs.heads[j].acc = f.acc, fs.heads[j].state = f.state;
    }
}

struct PathState {
    V3 orig, dir;   // the ray to cast next
    float t_exit;   // shadow ray: early-exit distance
    int light;      // < 0: the ray is a closest-hit ray; >= 0: shadow ray towards this light
    int depth;      // remaining depth of the node the ray belongs to
    int fullDepth;  // depth of the root (statistics only)
    int sp;         // frames in use
    // the node being shaded (valid while light >= 0)
    V3 point, N, view;
    Material m;
    float ndl, lightDist, Ld, Ls;
    int cast_mask;  // quad mode: which lights of the current batch of four get a shadow ray
    bool ahead;     // quad mode with look-ahead: lane 3 casts the first child's closest-hit ray beside the shadow rays
};

RTX_DEV void pathStart(PathState &st, V3 orig, V3 dir, int depth) {
    st.orig = orig, st.dir = dir, st.depth = depth, st.fullDepth = depth, st.sp = 0;
    st.light = -1, st.t_exit = -1.0f, st.cast_mask = 0, st.ahead = false;
}

// Quad mode.  Four neighbouring lanes (a "quad") can share ONE pixel: all four hold the same path
// state and cast the same closest-hit rays, but the shadow rays of a node go out four lights at a
// time, one per lane, and the verdicts are exchanged with DPP.  A node then takes two casts instead
// of one plus one per light, which halves the longest chains of the frame at the price of three
// redundant lanes; the kernel uses it for the few tiles that last longest.  quad_j is the lane's
// place in its quad (0..3), or -1 in the normal one-pixel-per-lane mode.
//
// Look-ahead (LOOK: the quad-mode launch of k_render).  With three lights the fourth lane of a quad has nothing to
// do in the shadow trip.  Where the node's first child ray goes depends on the hit alone, not on the lights
// (src/Scene.cpp:95-101), so that lane casts the child's closest-hit ray in the same trip; its hit record is handed
// round the quad afterwards and the child becomes the current node without a trip of its own.  A node then costs one
// trip instead of two.  The rays cast, their order per pixel and every sum are the same as without it.

// Does the pending ray have to be cast at all?  (A closest-hit ray of depth 0 is answered by the
// sky; in quad mode a lane casts only its own light of the batch.)
RTX_DEV bool pathCasts(const PathState &st, int quad_j) {
    if (st.light >= 0) return quad_j < 0 || ((st.cast_mask >> quad_j) & 1);
    return st.depth > 0;
}

// is this lane's pending ray the look-ahead ray of its quad (a closest-hit ray although st.light >= 0)?
RTX_DEV bool pathRayIsAhead(const PathState &st, int quad_j) { return st.ahead && quad_j == 3; }

template <bool STATS>
RTX_DEV void pathCountRay(const PathState &st, int quad_j, LaneCounters<STATS> &lc) {
    if constexpr (STATS) {
        if (pathRayIsAhead(st, quad_j)) {  // the child's closest-hit ray: it is not cast again when the child's turn comes
            lc.v[CNT_SECONDARY] += 1u;
            return;
        }
        const bool shadow = st.light >= 0;
        const bool first = quad_j <= 0;  // the four lanes of a quad cast the same closest-hit ray: count it once
        lc.v[CNT_SHADOW] += shadow ? 1u : 0u;
        lc.v[CNT_PRIMARY] += !shadow && first && st.depth == st.fullDepth ? 1u : 0u;
        lc.v[CNT_SECONDARY] += !shadow && first && st.depth != st.fullDepth ? 1u : 0u;
    }
}

// One light of calcLightning, src/Scene.cpp:64-79, in three pieces.
struct LightAim {
    V3 pos, dir;  // the light, the unit vector towards it
    float dist;   // distance(light, point), src/Scene.cpp:70
    float ndl, intensity;
    bool counts;  // false: the light can add to neither sum, its shadow ray decides nothing
};

template <bool WIDE = false>
RTX_DEV LightAim aimLight(const DevScene &s, const SmallScene &q, const PathState &st, int li) {
    LightAim a;
    const float4 L = lightAt<WIDE>(s, q, li);
    a.pos = v3(L.x, L.y, L.z);
    a.intensity = L.w;
    // normalize(light - point) and distance(light, point) share their square root: the reference forms the second from
    // point - light, whose components are the negated ones, and the squares of those are the same numbers
    const V3 to_light = a.pos - st.point;
    a.dist = sqrtf(dot(to_light, to_light));
    a.dir = to_light * (1.0f / a.dist);
    a.ndl = dot(a.dir, st.N);
    a.counts = true;
    if (s.prune) {
        // A light adds to the diffuse sum only if n.l >= 0 and to the specular sum only if the
        // mirrored direction faces the viewer (:75-78).  If it can add to neither, its shadow ray
        // decides nothing and is not cast.  The diffuse sum itself is irrelevant when the
        // material's diffuse weight is exactly 0 (x * Ld * 0 has the same value for every finite
        // Ld >= 0, which holds when no light has a negative or non-finite intensity).  The
        // specular term itself can be exactly 0 (a high exponent underflows quickly); adding +0 to
        // a sum of non-negative terms changes nothing either.
        const bool diffuseCounts = a.ndl >= 0 && !(st.m.a0 == 0.0f && s.lights_plain);
        if (!diffuseCounts) {
            const float specIntDot = dot(-reflect(-a.dir, st.N), st.view);
            bool specCounts = specIntDot >= 0;
            if (specCounts && s.lights_plain) specCounts = !(refPowf(q.math, specIntDot, st.m.spec) * a.intensity == 0.0f);
            a.counts = specCounts;
        }
    }
    return a;
}

// the shadow ray towards an aimed light becomes the pending ray, :65-70
RTX_DEV void pathAimShadowRay(const DevScene &s, PathState &st, const LightAim &a) {
#pragma clang fp contract(off)
    st.ndl = a.ndl;
    st.orig = st.point + st.N * (__builtin_signbit(a.ndl) ? -s.fprec : s.fprec);  // (= point - N * fprec bit for bit on the far side)
    st.dir = a.dir;
    st.lightDist = a.dist;
    // A hit nearer than this decides the reference's test for certain: the closest hit is then at
    // least that near, and |orig + dir*t - orig| differs from t by a few ulp only.
    st.t_exit = fminf(st.lightDist * 0.999f, s.far_clip);
}

// :71-72: is the pending shadow ray's light hidden?
RTX_DEV bool pathShadowed(const PathState &st, bool hit, float md) {
    if (!hit) return false;
    if (md < st.t_exit) return true;
    const V3 shadowPt = st.orig + st.dir * md;
    return distance(shadowPt, st.orig) < st.lightDist;
}

// :74-78: an unhidden light's share of the two sums
RTX_DEV void pathAddLight(const SmallScene &q, PathState &st, V3 lightDir, float ndl, float intensity) {
    if (ndl >= 0) st.Ld += intensity * ndl;
    const float specIntDot = dot(-reflect(-lightDir, st.N), st.view);
    if (specIntDot >= 0) st.Ls += refPowf(q.math, specIntDot, st.m.spec) * intensity;
}

// value of lane k of the caller's quad
template <int k>
RTX_DEV int quadBroadcast(int v) {
    return __builtin_amdgcn_update_dpp(0, v, k | (k << 2) | (k << 4) | (k << 6), 0xf, 0xf, true);
}

// Consumes the result of the pending ray (hit, h, md: what castEnd returned; hit = false when the ray
// was not cast) and sets up the next one.  Returns true when the whole tree is done; `color` is
// then the value traceRay would have returned.
// value of lane 3 of the caller's quad, for every field of a hit record
RTX_DEV float quadFromLane3(float v) { return __int_as_float(quadBroadcast<3>(__float_as_int(v))); }
RTX_DEV Hit quadHitFromLane3(const Hit &h) {
    Hit r;
    r.point = v3(quadFromLane3(h.point.x), quadFromLane3(h.point.y), quadFromLane3(h.point.z));
    r.N = v3(quadFromLane3(h.N.x), quadFromLane3(h.N.y), quadFromLane3(h.N.z));
    r.m.refractive = quadFromLane3(h.m.refractive), r.m.a0 = quadFromLane3(h.m.a0), r.m.a1 = quadFromLane3(h.m.a1);
    r.m.a2 = quadFromLane3(h.m.a2), r.m.a3 = quadFromLane3(h.m.a3);
    r.m.dr = quadFromLane3(h.m.dr), r.m.dg = quadFromLane3(h.m.dg), r.m.db = quadFromLane3(h.m.db), r.m.spec = quadFromLane3(h.m.spec);
    r.mat_index = quadBroadcast<3>(h.mat_index);
    return r;
}

// the origin of a node's child ray of direction d: the point moved along N to the side the child leaves on.
// point - N * e is point + N * (-e) bit for bit, which saves choosing between two vectors
RTX_DEV V3 pathChildOrigin(const DevScene &s, const PathState &st, V3 d) {
#pragma clang fp contract(off)
    return st.point + st.N * (__builtin_signbit(dot(d, st.N)) ? -s.fprec : s.fprec);
}

template <bool STATS, bool WIDE = false, bool LOOK = false>
RTX_DEV bool pathConsume(const DevScene &s, const SmallScene &q, PathState &st, FrameStack &frames, bool hit, const Hit &h_cast, float md,
                         int quad_j, V3 &color, LaneCounters<STATS> &lc) {
    const bool count = quad_j <= 0;  // statistics: once per pixel
    Hit h = h_cast;
    bool ahead_known = false, ahead_hit = false;  // LOOK: the first child's closest-hit result came with this trip
    Hit ahead_h = h_cast;
    for (;;) {  // (a second turn only with LOOK: for the child whose result is already here)
    const bool shadow = st.light >= 0;
    bool returning = false;          // `color` holds the finished value of a node
    if (shadow && quad_j < 0) {
        // ---- after the shadow ray of light st.light ----
        if (!pathShadowed(st, hit, md)) {
            RTX_COUNT(CNT_LIT);
            pathAddLight(q, st, st.dir, st.ndl, lightAt<WIDE>(s, q, st.light).w);
        }
        ++st.light;
    } else if (shadow) {
        // ---- quad mode: after the shadow rays of lights st.light .. st.light + 3, one per lane.  Every
        // lane of the quad adds all of them up, in the reference's order, so the four states stay equal.
        const int mine = ((st.cast_mask >> quad_j) & 1) && pathShadowed(st, hit, md) ? 1 : 0;
        const int hidden[4] = {quadBroadcast<0>(mine), quadBroadcast<1>(mine), quadBroadcast<2>(mine), quadBroadcast<3>(mine)};
        if constexpr (LOOK) {
            if (st.ahead) {  // lane 3 cast the first child's closest-hit ray: everyone takes its result
                ahead_known = true;
                ahead_hit = quadBroadcast<3>(hit ? 1 : 0) != 0;
                ahead_h = quadHitFromLane3(h);
                st.ahead = false;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int li = st.light + k;
            if (li < s.n_lights && ((st.cast_mask >> k) & 1) && !hidden[k]) {
                if constexpr (STATS) lc.v[CNT_LIT] += count ? 1u : 0u;
                const float4 L = lightAt<WIDE>(s, q, li);
                const V3 lightDir = normalize(v3(L.x, L.y, L.z) - st.point);
                pathAddLight(q, st, lightDir, dot(lightDir, st.N), L.w);
            }
        }
        st.light += 4;
    } else if (!hit) {
        if constexpr (STATS) lc.v[CNT_SKY] += count ? 1u : 0u;
        color = skyColor(s, st.dir);
        returning = true;
    } else {
        st.point = h.point, st.N = h.N, st.m = h.m, st.view = st.dir;
        st.Ld = 0.0f, st.Ls = 0.0f;
        st.light = 0;
    }
    if (!returning) {
        // ---- the next shadow ray(s) ----
        while (st.light < s.n_lights) {
            if (quad_j < 0) {
                const LightAim a = aimLight<WIDE>(s, q, st, st.light);
                if (!a.counts) {
                    ++st.light;
                    continue;
                }
                pathAimShadowRay(s, st, a);
                return false;
            }
            int mask = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (st.light + k < s.n_lights) {
                    const LightAim a = aimLight<WIDE>(s, q, st, st.light + k);
                    if (a.counts) {
                        mask |= 1 << k;
                        if (k == quad_j) pathAimShadowRay(s, st, a);
                    }
                }
            }
            if (mask) {
                if constexpr (LOOK) {
                    // the last batch of lights and lane 3 is free: it casts the closest-hit ray of the node's first child
                    // (the ray the code below will make the pending one when the lights are done)
                    if (st.light + 3 >= s.n_lights && st.depth > 1) {
                        const bool doRefl = !s.prune || st.m.a2 != 0.0f, doRefr = !s.prune || st.m.a3 != 0.0f;
                        if (doRefl || doRefr) {
                            st.ahead = true;
                            mask |= 8;
                            if (quad_j == 3) {
                                const V3 d = doRefl ? reflectDir(st.view, st.N) : normalize(refractDir(st.view, st.N, st.m.refractive));
                                st.orig = pathChildOrigin(s, st, d), st.dir = d, st.t_exit = -1.0f;
                            }
                        }
                    }
                }
                st.cast_mask = mask;
                return false;
            }
            st.light += 4;
        }
        // ---- all lights done: the node's own colour and its children, :96-105 ----
        st.light = -1, st.t_exit = -1.0f;
        const Material &m = st.m;
        const V3 dir = st.view;
        const int depth = st.depth;
        const V3 base = v3(m.dr * st.Ld * m.a0, m.dg * st.Ld * m.a0, m.db * st.Ld * m.a0) + v3(st.Ls * m.a1, st.Ls * m.a1, st.Ls * m.a1);
        const bool doRefl = !s.prune || m.a2 != 0.0f;
        const bool doRefr = !s.prune || m.a3 != 0.0f;
        if (!doRefl && !doRefr) {
            color = base;
        } else {
            const auto childOrigin = [&](V3 d) { return pathChildOrigin(s, st, d); };
            V3 reflD = v3(0, 0, 0), refrD = v3(0, 0, 0);
            if (doRefl) reflD = reflectDir(dir, st.N);
            if (doRefr) refrD = normalize(refractDir(dir, st.N, m.refractive));
            if (depth == 1) {
                // the children have depth 0: traceRay returns the sky colour of their direction at
                // once (src/Scene.cpp:87-88), so they are evaluated in place
                V3 c = base;
                if (doRefl) {
                    if constexpr (STATS) lc.v[CNT_SKY] += count ? 1u : 0u;
                    c = c + skyColor(s, reflD) * m.a2;
                }
                if (doRefr) {
                    if constexpr (STATS) lc.v[CNT_SKY] += count ? 1u : 0u;
                    c = c + skyColor(s, refrD) * m.a3;
                }
                color = c;
            } else {
                FrameHead f;
                f.acc = base;
                f.a2 = m.a2, f.a3 = m.a3;
                f.state = (doRefl ? 1 : 0) | (doRefr ? 2 : 0) | (depth << 8);
                V3 refrO = v3(0, 0, 0);
                if (doRefr) refrO = childOrigin(refrD);
                if (doRefl) {
                    st.orig = childOrigin(reflD);
                    st.dir = reflD;
                } else {
                    st.orig = refrO, st.dir = refrD;
                    f.state = (f.state & ~2) | 4;
                }
                const int j = st.sp++;
                frameSetHead(frames, j, f);
                // the pending refract ray is only stored when there is one (most nodes have the reflect child alone)
                if (f.state & 2) frames.rays[j].orig = refrO, frames.rays[j].dir = refrD;
                st.depth = depth - 1;
                if constexpr (LOOK) {
                    if (ahead_known) {  // the pending ray has been cast already: on with its result
                        ahead_known = false;
                        hit = ahead_hit, h = ahead_h;
                        continue;
                    }
                }
                return false;
            }
        }
    }
    // ---- return `color` to the waiting frames ----
    while (st.sp > 0) {
        const int j = st.sp - 1;
        FrameHead f = frameHead(frames, j);
        if (f.state & 1) {  // reflect child returned
            f.acc = f.acc + color * f.a2;
            if (f.state & 2) {  // now the refract child
                f.state = (f.state & ~3) | 4;
                frameSetSum(frames, j, f);
                st.orig = frames.rays[j].orig, st.dir = frames.rays[j].dir;
                st.depth = (f.state >> 8) - 1;
                return false;
            }
            color = f.acc;
        } else {  // refract child returned
            color = f.acc + color * f.a3;
        }
        --st.sp;
    }
    return true;
    }
}

template <bool STATS, bool WIDE = false>
RTX_DEV V3 traceRay(const DevScene &s, const SmallScene &q, V3 orig, V3 dir, int depth, int *stack, int stride, LaneCounters<STATS> &lc) {
    PathState st;
    FrameSpill spill;  // all frames in scratch memory here, the rest of the state in registers
    FrameStack frames = frameStack(spill);
    pathStart(st, orig, dir, depth);
    V3 color = v3(0, 0, 0);
    for (;;) {
        bool hit = false;
        Hit h;
        float md = 0.0f;
        if (pathCasts(st, -1)) {
            pathCountRay<STATS>(st, -1, lc);
            hit = sceneIntersect<STATS, WIDE>(s, q, st.orig, st.dir, st.light >= 0, st.t_exit, h, md, stack, stride, lc);
        }
        if (pathConsume<STATS, WIDE>(s, q, st, frames, hit, h, md, -1, color, lc)) return color;
    }
}

}  // namespace rtx
