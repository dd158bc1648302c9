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

#define RTX_COUNT(slot) \
    if constexpr (STATS) lc.v[slot]++
#define RTX_COUNT_N(slot, n) \
    if constexpr (STATS) lc.v[slot] += (n)

// ---------------------------------------------------------------------------------------------
// Mesh: LBVH traversal.  Returns the closest front-facing hit under Triangle::rayIntersect with
// t < best_t (strict, as modelsIntersect's `curDist < modelsDist && curDist < minDist`,
// src/SceneObjects.cpp:365-366).  `stack` points at this lane's column of an LDS array whose rows
// are `stride` ints apart, so the 64 lanes of a wave hit 64 different banks.
// In SHADOW mode the walk stops at the first hit nearer than t_exit.
// ---------------------------------------------------------------------------------------------
template <bool SHADOW, bool STATS>
RTX_DEV bool meshIntersect(const DevScene &s, V3 orig, V3 dir, V3 inv, float &best_t, int &best_tri, float t_exit,
                           int *stack, int stride, LaneCounters<STATS> &lc) {
    // the gate: Model::boxIntersect on the padded vertex bounds, src/SceneObjects.cpp:356 -- the
    // cheap filter first, the reference's own slab test (six IEEE divisions) only for candidates
    if (!slabMaybe(s.mesh_lo, s.mesh_hi, orig, inv)) return false;
    RTX_COUNT(CNT_BOX_TESTS);
    float gate_near = 0.0f;
    if (!boxSlab(s.mesh_lo, s.mesh_hi, orig, dir, gate_near)) return false;
    RTX_COUNT(CNT_BOX_HITS);

    // Box tests only steer the walk; they use reciprocals and FMA freely.  Node boxes were padded
    // when the tree was built, and t_far is widened by 2 ulp, so a triangle the exact test would
    // accept is never culled by rounding here.
    const float ix = inv.x, iy = inv.y, iz = inv.z;  // clamped reciprocals: no inf - inf below
    const float ox = -orig.x * ix, oy = -orig.y * iy, oz = -orig.z * iz;
    int sp = 0;
    int node = 0;
    int best_key = 0x7fffffff;  // order in which the reference's cell walk reaches best_tri
    for (;;) {
        const float4 *np = reinterpret_cast<const float4 *>(s.bvh + node);
        const float4 a = np[0], b = np[1], c = np[2];
        const int4 ch = *reinterpret_cast<const int4 *>(np + 3);
        RTX_COUNT(CNT_NODE_TESTS);
        float tnL, tfL, tnR, tfR;
        {
            float t0 = __builtin_fmaf(a.x, ix, ox), t1 = __builtin_fmaf(a.w, ix, ox);
            float t2 = __builtin_fmaf(a.y, iy, oy), t3 = __builtin_fmaf(b.x, iy, oy);
            float t4 = __builtin_fmaf(a.z, iz, oz), t5 = __builtin_fmaf(b.y, iz, oz);
            tnL = fmaxf(fmaxf(fminf(t0, t1), fminf(t2, t3)), fmaxf(fminf(t4, t5), 0.0f));
            tfL = fminf(fminf(fmaxf(t0, t1), fmaxf(t2, t3)), fminf(fmaxf(t4, t5), best_t)) * 1.0000004f;
        }
        {
            float t0 = __builtin_fmaf(b.z, ix, ox), t1 = __builtin_fmaf(c.y, ix, ox);
            float t2 = __builtin_fmaf(b.w, iy, oy), t3 = __builtin_fmaf(c.z, iy, oy);
            float t4 = __builtin_fmaf(c.x, iz, oz), t5 = __builtin_fmaf(c.w, iz, oz);
            tnR = fmaxf(fmaxf(fminf(t0, t1), fminf(t2, t3)), fmaxf(fminf(t4, t5), 0.0f));
            tfR = fminf(fminf(fmaxf(t0, t1), fmaxf(t2, t3)), fminf(fmaxf(t4, t5), best_t)) * 1.0000004f;
        }
        bool hitL = tnL <= tfL, hitR = tnR <= tfR;
        int next = -1;  // inner node to descend into
        int defer = -1; // inner node to push
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const bool h = side == 0 ? hitL : hitR;
            const int child = side == 0 ? ch.x : ch.y;
            if (h && child < 0) {  // leaf: one triangle
                const int ti = ~child;
                const float4 *tp = s.mesh_tris + 3 * (size_t)ti;
                const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
                RTX_COUNT(CNT_TRI_TESTS);
                float t = 0.0f;
                bool past_u = false;
                const bool th = triHit(v3(p0.x, p0.y, p0.z), v3(p1.x, p1.y, p1.z), v3(p2.x, p2.y, p2.z), orig, dir, t,
                                       STATS ? &past_u : nullptr);
                if constexpr (STATS) lc.v[CNT_TRI_FULL] += past_u ? 1u : 0u;
                if (th && t <= best_t) {
                    // of two hits at the same distance the reference keeps the one it tests first
                    int key = ti;
                    if (s.mesh_cells) {
                        // the triangle exists for this ray only if the ray passes a cell holding one
                        // of its vertices; the i,j,k cell walk reaches it in the order of the keys
                        const int4 keys = s.mesh_cells[ti];
                        key = 0x7fffffff;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const int k = c == 0 ? keys.x : (c == 1 ? keys.y : keys.z);
                            if (k < 0 || k >= key) continue;
                            const float *cb = s.grid_cells + 6 * (k >> 22);
                            const float lo[3] = {cb[0], cb[1], cb[2]}, hi[3] = {cb[3], cb[4], cb[5]};
                            float cell_near = 0.0f;
                            if (slabMaybe(lo, hi, orig, inv) && boxSlab(lo, hi, orig, dir, cell_near)) key = k;
                        }
                    }
                    const bool take = key != 0x7fffffff && (t < best_t || (best_tri >= 0 && key < best_key));
                    if (take) best_key = key;
                    if (take) {
                        best_t = t;
                        best_tri = ti;
                        if (SHADOW && t < t_exit) return true;
                    }
                }
            }
        }
        const bool innerL = hitL && ch.x >= 0, innerR = hitR && ch.y >= 0;
        if (innerL && innerR) {
            const bool leftFirst = tnL <= tnR;
            next = leftFirst ? ch.x : ch.y;
            defer = leftFirst ? ch.y : ch.x;
        } else if (innerL) {
            next = ch.x;
        } else if (innerR) {
            next = ch.y;
        }
        if (defer >= 0) {
            stack[sp * stride] = defer;
            ++sp;
        }
        if (next < 0) {
            if (sp == 0) break;
            --sp;
            next = stack[sp * stride];
        }
        node = next;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// Scene::sceneIntersect, src/Scene.cpp:29-58: categories in the reference's order, each accepting
// only a strictly nearer hit.
//   SHADOW = false: fills h (point, normal, material copy with water / island colour override).
//   SHADOW = true : only the distance matters; returns as soon as any hit nearer than t_exit is
//                   known (the caller chooses t_exit so that the reference's own test
//                   `distance(shadowPt, shadowOrig) < distance(light, point)` is then certain).
// Returns `minDist < far_clip`; minDist itself comes back in min_dist.
// ---------------------------------------------------------------------------------------------
template <bool SHADOW, bool STATS>
RTX_DEV bool sceneIntersect(const DevScene &s, const SmallScene &q, V3 orig, V3 dir, Hit &h, float &min_dist, float t_exit, int *stack,
                            int stride, LaneCounters<STATS> &lc) {
    float minDist = kFltMax;
    // what was hit last: 0 none, 1 sphere, 2 loose triangle, 3 box, 4 mesh triangle, 5 water, 6 island
    int kind = 0, index = -1;
    const V3 inv = safeInverse(dir);

    // spheresIntersect, src/SceneObjects.cpp:289-302
    if (s.n_spheres > 0 && slabMaybe(s.sphere_bounds, s.sphere_bounds + 3, orig, inv)) {
        RTX_COUNT_N(CNT_SPHERE_TESTS, s.n_spheres);
        for (int i = 0; i < s.n_spheres; ++i) {
            const float4 sp = q.spheres[i];
            float curDist = 0.0f;
            if (sphereHit(v3(sp.x, sp.y, sp.z), sp.w, orig, dir, curDist) && curDist < minDist) {
                minDist = curDist;
                kind = 1, index = i;
            }
        }
        if (SHADOW && minDist < t_exit) {
            min_dist = minDist;
            return true;
        }
    }
    // trianglesIntersect, :304-317
    if (s.n_tris > 0 && slabMaybe(s.tri_bounds, s.tri_bounds + 3, orig, inv)) {
        RTX_COUNT_N(CNT_LOOSE_TRI_TESTS, s.n_tris);
        for (int i = 0; i < s.n_tris; ++i) {
            const float *t = q.tris + 9 * i;
            const V3 v0 = v3(t[0], t[1], t[2]), v1 = v3(t[3], t[4], t[5]), v2 = v3(t[6], t[7], t[8]);
            float curDist = 0.0f;
            if (triHit(v0, v1 - v0, v2 - v0, orig, dir, curDist) && curDist < minDist) {
                minDist = curDist;
                kind = 2, index = i;
            }
        }
    }
    // cubesIntersect, :382-397
    if (s.n_boxes > 0 && slabMaybe(s.box_bounds, s.box_bounds + 3, orig, inv)) {
        for (int i = 0; i < s.n_boxes; ++i) {
            const float *b = q.boxes + 6 * i;
            const float lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[3], b[4], b[5]};
            if (!slabMaybe(lo, hi, orig, inv)) continue;
            RTX_COUNT(CNT_BOX_TESTS);
            float curDist = 0.0f;
            if (boxSlab(lo, hi, orig, dir, curDist) && curDist < minDist) {
                minDist = curDist;
                kind = 3, index = i;
            }
        }
        if (SHADOW && minDist < t_exit) {
            min_dist = minDist;
            return true;
        }
    }
    // modelsIntersect, :350-380
    if (s.n_mesh_tris > 0) {
        int tri = -1;
        float t = minDist;
        if (meshIntersect<SHADOW, STATS>(s, orig, dir, inv, t, tri, t_exit, stack, stride, lc)) {
            min_dist = t;
            return true;
        }
        if (tri >= 0) {
            minDist = t;
            kind = 4, index = tri;
        }
    }
    // water plane, src/Scene.cpp:38-54
    if (s.water_w > 0) {
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
    {
        float islandsDist = kFltMax;
        float dist = 0.0f;
        RTX_COUNT_N(CNT_PLANE_TESTS, s.n_islands > 0 ? 1 : 0);
        if (s.n_islands > 0 && planeHit(s.island_y, orig, dir, dist)) {
            RTX_COUNT_N(CNT_ISLAND_TESTS, s.n_islands);
            const V3 pt = orig + dir * dist;
            for (int i = 0; i < s.n_islands; ++i) {
                const float4 is = q.islands[i];  // cx, cz, a, b
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
    if (SHADOW) return minDist < s.far_clip;

    // ---- resolve the winner: hit point, normal, material ----
    if (kind == 0) return false;
    h.point = orig + dir * minDist;
    int mat = 0;
    if (kind == 1) {
        const float4 sp = q.spheres[index];
        h.N = normalize(h.point - v3(sp.x, sp.y, sp.z));
        mat = q.sphere_mat[index];
    } else if (kind == 2) {
        const float *t = q.tris + 9 * index;
        const V3 v0 = v3(t[0], t[1], t[2]), v1 = v3(t[3], t[4], t[5]), v2 = v3(t[6], t[7], t[8]);
        h.N = triNormal(v1 - v0, v2 - v0);
        mat = q.tri_mat[index];
    } else if (kind == 3) {
        const float *b = q.boxes + 6 * index;
        const float lo[3] = {b[0], b[1], b[2]}, hi[3] = {b[3], b[4], b[5]};
        h.N = boxNormal(lo, hi, h.point);
        mat = q.box_mat[index];
    } else if (kind == 4) {
        const float4 *tp = s.mesh_tris + 3 * (size_t)index;
        const float4 p1 = tp[1], p2 = tp[2];
        h.N = triNormal(v3(p1.x, p1.y, p1.z), v3(p2.x, p2.y, p2.z));
        mat = s.mesh_mat;
    } else if (kind == 5) {
        h.N = v3(0.0f, 1.0f, 0.0f);
        mat = s.water_mat;
    } else {
        h.N = v3(0.0f, 1.0f, 0.0f);
        mat = s.island_mat;
    }
    h.m = q.materials[mat];
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
        const float *c0 = parity ? s.light_green : s.dark_green;
        const float *c1 = parity ? s.dark_green : s.light_green;
        float x = h.point.x, fx = roundf(h.point.x);
        float z = ceilf(h.point.x + h.point.z) - fx - 0.5f;
        const float rad = 0.15f;
        const bool dot_ = (x - fx) * (x - fx) + (h.point.z - z) * (h.point.z - z) < rad * rad;
        const float *c = dot_ ? c1 : c0;
        h.m.dr = c[0], h.m.dg = c[1], h.m.db = c[2];
    }
    return minDist < s.far_clip;
}

// Sky::getColorAt, src/SceneObjects.cpp:440-445.  atan2 and acos bind to the float overloads in
// that translation unit; the two quotients are then formed in double and truncated to int; the
// row shift wraps in uint32 before the modulo.  Texel = byte * (1/255.0f) (:429-434).
RTX_DEV V3 skyColor(const DevScene &s, V3 dir) {
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

// Scene::calcLightning, src/Scene.cpp:60-81
template <bool STATS>
RTX_DEV void calcLighting(const DevScene &s, const SmallScene &q, V3 dir, V3 point, V3 N, float specular_exp, float &Ld, float &Ls,
                          int *stack, int stride, LaneCounters<STATS> &lc) {
    Ld = 0.0f, Ls = 0.0f;
    const V3 positiveOrig = point + N * s.fprec, negativeOrig = point - N * s.fprec;
    for (int i = 0; i < s.n_lights; ++i) {
        const float4 L = q.lights[i];
        const V3 lpos = v3(L.x, L.y, L.z);
        const V3 lightDir = normalize(lpos - point);
        const float ndl = dot(lightDir, N);
        const V3 shadowOrig = __builtin_signbit(ndl) ? negativeOrig : positiveOrig;
        const float lightDist = distance(lpos, point);
        // A hit nearer than this decides the reference's test for certain: the closest hit is
        // then at least that near, and |orig + dir*t - orig| differs from t by a few ulp only.
        const float t_exit = fminf(lightDist * 0.999f, s.far_clip);
        Hit dummy;
        float minDist;
        RTX_COUNT(CNT_SHADOW);
        bool occluded = false;
        if (sceneIntersect<true, STATS>(s, q, shadowOrig, lightDir, dummy, minDist, t_exit, stack, stride, lc)) {
            if (minDist < t_exit) {
                occluded = true;
            } else {
                const V3 shadowPt = shadowOrig + lightDir * minDist;
                occluded = distance(shadowPt, shadowOrig) < lightDist;
            }
        }
        if (occluded) continue;
        RTX_COUNT(CNT_LIT);
        if (ndl >= 0) Ld += L.w * ndl;
        const float specIntDot = dot(-reflect(-lightDir, N), dir);
        if (specIntDot >= 0) Ls += refPowf(q.math, specIntDot, specular_exp) * L.w;
    }
}

// ---------------------------------------------------------------------------------------------
// Scene::traceRay, src/Scene.cpp:83-106, without recursion.  A frame is pushed for every hit
// that has at least one child to wait for; children are traced reflect first, refract second, and
// the sum is formed left to right exactly as at :103-105:
//     ((diffuse*Ld*a0 + 1*Ls*a1) + reflect*a2) + refract*a3.
// With pruning on, a child whose weight is exactly 0 is not traced and its (exactly zero)
// contribution not added; x + 0*c == x for every finite c, so the result is unchanged.
// ---------------------------------------------------------------------------------------------
struct Frame {
    V3 acc;          // partial sum so far
    float a2, a3;
    V3 rorig, rdir;  // pending refract ray
    int state;       // bit0: waiting for reflect, bit1: refract still to do, bit2: waiting for refract, bits 8..: depth of this node
};

// One path = one traceRay call tree, walked one node per step so that a wave can interleave the
// paths of different pixels (lanes that finish take a new pixel while their neighbours go deeper).
struct PathState {
    V3 orig, dir;
    int depth;      // remaining depth of the node to evaluate next
    int fullDepth;  // depth of the root (statistics only)
    int sp;         // frames in use
    Frame frames[kMaxDepth];
};

RTX_DEV void pathStart(PathState &st, V3 orig, V3 dir, int depth) {
    st.orig = orig, st.dir = dir, st.depth = depth, st.fullDepth = depth, st.sp = 0;
}

// Evaluates the next node of the path.  Returns true when the whole tree is done; `color` is then
// the value traceRay would have returned.
template <bool STATS>
RTX_DEV bool pathStep(const DevScene &s, const SmallScene &q, PathState &st, V3 &color, int *stack, int stride, LaneCounters<STATS> &lc) {
    const V3 orig = st.orig, dir = st.dir;
    const int depth = st.depth;
    bool hit = false;
    Hit h;
    if (depth > 0) {
        if constexpr (STATS) lc.v[depth == st.fullDepth ? CNT_PRIMARY : CNT_SECONDARY]++;
        float md;
        hit = sceneIntersect<false, STATS>(s, q, orig, dir, h, md, 0.0f, stack, stride, lc);
    }
    if (!hit) {
        RTX_COUNT(CNT_SKY);
        color = skyColor(s, dir);
    } else {
        float Ld, Ls;
        calcLighting<STATS>(s, q, dir, h.point, h.N, h.m.spec, Ld, Ls, stack, stride, lc);
        const V3 base = v3(h.m.dr * Ld * h.m.a0, h.m.dg * Ld * h.m.a0, h.m.db * Ld * h.m.a0) +
                        v3(Ls * h.m.a1, Ls * h.m.a1, Ls * h.m.a1);
        const bool doRefl = !s.prune || h.m.a2 != 0.0f;
        const bool doRefr = !s.prune || h.m.a3 != 0.0f;
        if (!doRefl && !doRefr) {
            color = base;
        } else {
            const V3 positiveOrig = h.point + h.N * s.fprec, negativeOrig = h.point - h.N * s.fprec;
            V3 reflD = v3(0, 0, 0), refrD = v3(0, 0, 0);
            if (doRefl) reflD = normalize(reflect(dir, h.N));
            if (doRefr) refrD = normalize(refractDir(dir, h.N, h.m.refractive));
            if (depth == 1) {
                // the children have depth 0: traceRay returns the sky colour of their direction at
                // once (src/Scene.cpp:87-88), so they are evaluated in place
                V3 c = base;
                if (doRefl) {
                    RTX_COUNT(CNT_SKY);
                    c = c + skyColor(s, reflD) * h.m.a2;
                }
                if (doRefr) {
                    RTX_COUNT(CNT_SKY);
                    c = c + skyColor(s, refrD) * h.m.a3;
                }
                color = c;
            } else {
                Frame f;
                f.acc = base;
                f.a2 = h.m.a2, f.a3 = h.m.a3;
                f.state = (doRefl ? 1 : 0) | (doRefr ? 2 : 0) | (depth << 8);
                f.rdir = refrD;
                f.rorig = __builtin_signbit(dot(refrD, h.N)) ? negativeOrig : positiveOrig;
                if (doRefl) {
                    st.orig = __builtin_signbit(dot(reflD, h.N)) ? negativeOrig : positiveOrig;
                    st.dir = reflD;
                } else {
                    st.orig = f.rorig, st.dir = f.rdir;
                    f.state = (f.state & ~2) | 4;
                }
                st.frames[st.sp++] = f;
                st.depth = depth - 1;
                return false;
            }
        }
    }
    // ---- return `color` to the waiting frames ----
    while (st.sp > 0) {
        Frame &f = st.frames[st.sp - 1];
        if (f.state & 1) {  // reflect child returned
            f.acc = f.acc + color * f.a2;
            f.state &= ~1;
            if (f.state & 2) {  // now the refract child
                f.state = (f.state & ~2) | 4;
                st.orig = f.rorig, st.dir = f.rdir;
                st.depth = (f.state >> 8) - 1;
                return false;
            }
            color = f.acc;
            --st.sp;
        } else {  // refract child returned
            color = f.acc + color * f.a3;
            --st.sp;
        }
    }
    return true;
}

template <bool STATS>
RTX_DEV V3 traceRay(const DevScene &s, const SmallScene &q, V3 orig, V3 dir, int depth, int *stack, int stride, LaneCounters<STATS> &lc) {
    PathState st;
    pathStart(st, orig, dir, depth);
    V3 color = v3(0, 0, 0);
    while (!pathStep<STATS>(s, q, st, color, stack, stride, lc)) {
    }
    return color;
}

}  // namespace rtx
