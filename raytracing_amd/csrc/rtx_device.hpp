// Device-side scene layout and the per-ray routines of the trace loop (gfx950, wave64).
//
// Arithmetic policy.  Every value that can reach a pixel is computed in IEEE binary32 in the
// reference's operation order with NO fused multiply-add (this translation unit is built with
// -ffp-contract=off and the routines below never call fmaf), because the reference is an x86-64
// SSE2 build and a contracted a*b+c differs from it in the last bit.  The only places that use
// FMA on purpose are the BVH box tests, which decide what gets tested, never what a test returns.
//
// Reference citations are relative to the reference checkout (src/..., include/...).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtx_reftables.hpp"

namespace rtx {

#define RTX_DEV __device__ __forceinline__

// Everything in this file is geometry -- what decides which texel, which side of a shadow's edge, which way a mirrored
// ray goes -- and is never contracted, in the tolerance build either (its compiler flags allow fused multiply-adds; the
// pragma below takes that back until the end of the file).  A direction that is off by an ulp looks up another texel of
// the environment map in a mirror ball, and the neighbouring texel differs by far more than 1/255.
#pragma clang fp contract(off)

constexpr float kEps = 1.1920928955078125e-07f;   // std::numeric_limits<float>::epsilon()
constexpr float kFltMax = 3.4028234663852886e+38f;
constexpr float kFltMin = 1.1754943508222875e-38f;  // numeric_limits<float>::min()
constexpr int kMaxDepth = 8;                        // RTX_MAX_DEPTH

struct V3 {
    float x, y, z;
};
RTX_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RTX_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTX_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RTX_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RTX_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RTX_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// GLM's dot: tmp = a*b; tmp.x + tmp.y + tmp.z (include/glm/detail/func_geometric.inl:48-55)
RTX_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// func_geometric.inl:68-79
RTX_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
// v * (1 / sqrt(dot(v, v))): func_geometric.inl:82-90, func_exponential.inl:44-50
RTX_DEV V3 normalize(V3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
// I - N * dot(N, I) * 2: func_geometric.inl:104-110
RTX_DEV V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0f; }
RTX_DEV float distance(V3 p0, V3 p1) {
    V3 d = p1 - p0;
    return sqrtf(dot(d, d));
}

// ---------------------------------------------------------------------------------------------
// Scene as the kernels see it.  Everything is read-only during a render.  The struct is passed
// by value as a kernel argument, so its scalars and pointers sit in SGPRs and every loop over the
// analytic primitives reads them with scalar loads.
// ---------------------------------------------------------------------------------------------
struct Material {  // 9 floats, include/Material.h:8-18
    float refractive, a0, a1, a2, a3, dr, dg, db, spec;
};

// BVH2 node, 64 bytes: both children's boxes live in the parent so one fetch decides two subtrees.
// child >= 0: inner node index; child < 0: ~child is a triangle index.
struct BvhNode {
    float lminx, lminy, lminz, lmaxx;
    float lmaxy, lmaxz, rminx, rminy;
    float rminz, rmaxx, rmaxy, rmaxz;
    int left, right, pad0, pad1;
};

struct DevScene {
    // materials, lights
    const Material *materials;
    int n_materials;
    const float4 *lights;  // xyz, intensity
    int n_lights;
    // spheres: xyz, radius; material index
    const float4 *spheres;
    const int *sphere_mat;
    int n_spheres;
    // More spheres than the LDS staging holds (kMaxSmallPrims): a BVH2 over their boxes (leaf ~child =
    // sphere index), null otherwise.  The boxes are the geometric ones; the slack that the rounding
    // of Sphere::rayIntersect needs is added per ray (sphereWalk in rtx_trace.hpp).
    const BvhNode *sphere_bvh;
    float sphere_root_c[3];  // centre of the box around all spheres
    float sphere_root_h;     // its half diagonal
    float sphere_rmax;       // largest radius
    // loose triangles: v0, v1, v2 (9 floats each)
    const float *tris;
    const int *tri_mat;
    int n_tris;
    // boxes: lo xyz, hi xyz (6 floats each)
    const float *boxes;
    const int *box_mat;
    int n_boxes;
    // mesh: triangles as v0, e1 = v1 - v0, e2 = v2 - v0 in three float4 (w unused), BVH2 over them
    const float4 *mesh_tris;
    const BvhNode *bvh;
    int n_mesh_tris;
    int mesh_mat;
    float mesh_lo[3], mesh_hi[3];  // the gate box: all vertices -/+ 1e-4
    // The reference bins every mesh triangle into the cells of a 7x7x7 grid that hold its vertices
    // and tests it only for rays that pass one of those cells (src/SceneObjects.cpp:176-209,
    // 350-380).  mesh_cells[t] = up to three keys (cell index << 22 | position in the cell's list,
    // -1 = unused), null when that filter is switched off; cell (i, j, k), index (i*7 + j)*7 + k,
    // spans mesh_lo + grid_step * (i, j, k) .. mesh_lo + grid_step * (i+1, j+1, k+1).
    const int4 *mesh_cells;
    float grid_step[3];
    float grid_inv_step[3];  // approximate 1 / grid_step, for the hit-point shortcut only
    float grid_fast_tmax;    // hits beyond this distance never take the shortcut (0 disables it)
    // conservative bounds of whole categories (lo xyz, hi xyz), padded: a ray that misses one
    // cannot hit anything inside, so the exact per-primitive tests are skipped for it
    float sphere_bounds[6], tri_bounds[6], box_bounds[6];
    float object_bounds[6];  // union of the three and the mesh gate box
    float island_rect[4];    // x min, x max, z min, z max over all islands, padded
    // islands
    const float4 *islands;  // cx, cz, a, b
    int n_islands;
    int island_mat;
    float island_y;
    float light_green[3], dark_green[3];
    // water
    const unsigned char *water;
    int water_w, water_h, water_mat;
    float water_y, water_scale;
    // sky
    const unsigned char *sky;
    int sky_w, sky_h, sky_shift;
    // constants
    float far_clip;  // 700, src/Scene.cpp:57
    float fprec;     // 1e-3, include/Scene.h:23
    int prune;       // skip rays whose contribution is exactly nothing (zero-weight children, lights that cannot add)
    int lights_plain;  // every light intensity is finite and >= 0
};

// The analytic part of the scene (a few hundred bytes to a few KB) staged in LDS once per block.
// Every ray of every lane walks these lists; from global memory each list element costs a
// dependent scalar-load round trip per ray, from LDS a broadcast read.  The staging holds the first
// kMaxMaterials / kMaxLights / kMaxSmallPrims entries of each list; longer lists are legal, their
// tails are read from global memory (the loops run over the staged part first, then over the tail, so
// the reference's list order -- which decides ties -- is kept).
constexpr int kMaxMaterials = 32, kMaxLights = 8, kMaxSmallPrims = 16;
struct SmallScene {
    Material materials[kMaxMaterials];
    float4 lights[kMaxLights];
    float4 spheres[kMaxSmallPrims];
    float4 islands[kMaxSmallPrims];
    float tris[kMaxSmallPrims * 9];
    float boxes[kMaxSmallPrims * 6];
    // material index of whatever a ray can hit, in ONE array: spheres at 0, loose triangles at kMaxSmallPrims, boxes at
    // 2 kMaxSmallPrims, then the mesh, the water, the islands.  The hit record reads it with a computed slot; read from a
    // list per kind (and from kernel arguments for the last three) the compiler is free to merge the reads into one
    // flat load of a selected address, which waits for both memory pipes -- and did, on and off, with unrelated edits.
    int prim_mat[3 * kMaxSmallPrims + 4];
    // Constants of the scene that every ray looks at (copies of DevScene's): as kernel arguments they would sit in
    // scalar registers for the whole kernel, which has too few of those (the overflow is parked in vector-register
    // lanes and fetched back with a vector instruction per use); from here they cost a broadcast read.
    float sphere_bounds[6], tri_bounds[6], box_bounds[6], object_bounds[6];
    float island_rect[4];
    float light_green[3], dark_green[3];
    RefMathTables math;  // powf tables (rtx_refmath.hpp), indexed per lane
};

// cooperative copy by all threads of the block; the caller synchronises afterwards
__device__ __forceinline__ void loadSmallScene(SmallScene &q, const DevScene &s) {
    const int t = threadIdx.x, n = blockDim.x;
    auto copy = [&](void *dst, const void *src, int dwords) {
        for (int i = t; i < dwords; i += n) static_cast<unsigned *>(dst)[i] = static_cast<const unsigned *>(src)[i];
    };
    auto staged = [](int n_list, int room) { return n_list < room ? n_list : room; };
    copy(q.materials, s.materials, staged(s.n_materials, kMaxMaterials) * 9);
    copy(q.lights, s.lights, staged(s.n_lights, kMaxLights) * 4);
    if (s.n_spheres <= kMaxSmallPrims) {  // larger sets stay in global memory behind their BVH
        copy(q.spheres, s.spheres, s.n_spheres * 4);
        copy(q.prim_mat, s.sphere_mat, s.n_spheres);
    }
    copy(q.islands, s.islands, staged(s.n_islands, kMaxSmallPrims) * 4);
    copy(q.tris, s.tris, staged(s.n_tris, kMaxSmallPrims) * 9);
    copy(q.prim_mat + kMaxSmallPrims, s.tri_mat, staged(s.n_tris, kMaxSmallPrims));
    copy(q.boxes, s.boxes, staged(s.n_boxes, kMaxSmallPrims) * 6);
    copy(q.prim_mat + 2 * kMaxSmallPrims, s.box_mat, staged(s.n_boxes, kMaxSmallPrims));
    if (t == 0) q.prim_mat[3 * kMaxSmallPrims] = s.mesh_mat, q.prim_mat[3 * kMaxSmallPrims + 1] = s.water_mat, q.prim_mat[3 * kMaxSmallPrims + 2] = s.island_mat;
    copy(&q.math, &kRefMathTables, (int)(sizeof(RefMathTables) / 4));
    copy(q.sphere_bounds, s.sphere_bounds, 6), copy(q.tri_bounds, s.tri_bounds, 6), copy(q.box_bounds, s.box_bounds, 6);
    copy(q.object_bounds, s.object_bounds, 6);
    copy(q.island_rect, s.island_rect, 4);
    copy(q.light_green, s.light_green, 3), copy(q.dark_green, s.dark_green, 3);
}

// List entry i: from the staging while it reaches, from global memory beyond.  Only the general build of the
// kernels (WIDE) knows about lists longer than the staging; the host launches it when a scene has one.
// (The staged entry is read through a pointer that says LDS: through plain pointers the compiler merges the LDS read
// and the global read of such a pair into one flat load of a selected address, slower than either and a wait for both
// memory pipes.  Only the general build has the pair; in the default build every read below is an LDS read anyway.)
using LdsFloats = const __attribute__((address_space(3))) float *;
using LdsInts = const __attribute__((address_space(3))) int *;
__device__ __forceinline__ LdsFloats inLds(const float *p) { return (LdsFloats)p; }
__device__ __forceinline__ LdsInts inLds(const int *p) { return (LdsInts)p; }
__device__ __forceinline__ float4 ldsFloat4(const float4 *p) {
    const LdsFloats f = inLds(reinterpret_cast<const float *>(p));
    return make_float4(f[0], f[1], f[2], f[3]);
}
template <bool WIDE> __device__ __forceinline__ float4 lightAt(const DevScene &s, const SmallScene &q, int i) {
    if (WIDE && i >= kMaxLights) return s.lights[i];
    return ldsFloat4(q.lights + i);
}
template <bool WIDE> __device__ __forceinline__ Material materialAt(const DevScene &s, const SmallScene &q, int i) {
    if (WIDE && i >= kMaxMaterials) return s.materials[i];
    const LdsFloats f = inLds(reinterpret_cast<const float *>(q.materials + i));
    return Material{f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8]};
}
template <bool WIDE> __device__ __forceinline__ float4 islandAt(const DevScene &s, const SmallScene &q, int i) {
    if (WIDE && i >= kMaxSmallPrims) return s.islands[i];
    return ldsFloat4(q.islands + i);
}
// a loose triangle's nine floats, a box's six
struct Floats9 { float v[9]; __host__ __device__ const float &operator[](int k) const { return v[k]; } };
struct Floats6 { float v[6]; __host__ __device__ const float &operator[](int k) const { return v[k]; } };
template <bool WIDE> __device__ __forceinline__ Floats9 triAt(const DevScene &s, const SmallScene &q, int i) {
    Floats9 r;
    if (WIDE && i >= kMaxSmallPrims) {
        const float *g = s.tris + 9 * (size_t)i;
#pragma unroll
        for (int k = 0; k < 9; ++k) r.v[k] = g[k];
    } else {
        const LdsFloats f = inLds(q.tris + 9 * i);
#pragma unroll
        for (int k = 0; k < 9; ++k) r.v[k] = f[k];
    }
    return r;
}
template <bool WIDE> __device__ __forceinline__ Floats6 boxAt(const DevScene &s, const SmallScene &q, int i) {
    Floats6 r;
    if (WIDE && i >= kMaxSmallPrims) {
        const float *g = s.boxes + 6 * (size_t)i;
#pragma unroll
        for (int k = 0; k < 6; ++k) r.v[k] = g[k];
    } else {
        const LdsFloats f = inLds(q.boxes + 6 * i);
#pragma unroll
        for (int k = 0; k < 6; ++k) r.v[k] = f[k];
    }
    return r;
}
// material index of hit kind 1..6 (sphere, loose triangle, box, mesh triangle, water, island), list entry i
template <bool WIDE> __device__ __forceinline__ int hitMaterialIndex(const DevScene &s, const SmallScene &q, int kind, int i) {
    if constexpr (WIDE) {
        if (kind == 1 && s.sphere_bvh) return s.sphere_mat[i];
        if ((kind == 2 || kind == 3) && i >= kMaxSmallPrims) return (kind == 2 ? s.tri_mat : s.box_mat)[i];
    }
    return inLds(q.prim_mat)[kind <= 3 ? (kind - 1) * kMaxSmallPrims + i : 3 * kMaxSmallPrims + kind - 4];
}

// Statistics slots, in the order of rtx_stats' 64-bit fields (include/rtx.h).
enum CounterSlot {
    CNT_PRIMARY, CNT_SECONDARY, CNT_SHADOW, CNT_SKY, CNT_BOX_HITS, CNT_NODE_TESTS, CNT_TRI_TESTS, CNT_TRI_FULL,
    CNT_SPHERE_TESTS, CNT_LOOSE_TRI_TESTS, CNT_BOX_TESTS, CNT_PLANE_TESTS, CNT_ISLAND_TESTS, CNT_LIT, CNT_NODE_TRIPS, CNT_TRI_TRIPS, CNT_SLOTS = 16
};
struct Counters {
    unsigned long long v[CNT_SLOTS];
};

struct Hit {
    V3 point, N;
    Material m;
    int mat_index;  // m = materials[mat_index], diffuse overridden by the water texture / island pattern
};

// ---------------------------------------------------------------------------------------------
// Primitive tests
// ---------------------------------------------------------------------------------------------

// Sphere::rayIntersect, src/SceneObjects.cpp:15-26
RTX_DEV bool sphereHit(V3 center, float radius, V3 orig, V3 dir, float &dist) {
    V3 diff = center - orig;
    float t0 = dot(diff, dir);
    float dSquared = dot(diff, diff) - t0 * t0;
    float radius2 = radius * radius;
    if (dSquared > radius2) return false;
    float t1 = sqrtf(radius2 - dSquared);
    dist = t0 > t1 + kEps ? t0 - t1 : t0 + t1;
    return dist > kEps;
}

// Triangle::rayIntersect, src/SceneObjects.cpp:28-49, on precomputed edges (edge1 = v1 - v0 and
// edge2 = v2 - v0 are the same float subtractions the reference performs per call).
RTX_DEV bool triHit(V3 v0, V3 edge1, V3 edge2, V3 orig, V3 dir, float &dist, bool *past_u = nullptr) {
    V3 pvec = cross(dir, edge2);
    float det = dot(edge1, pvec);
    if (det < kEps) return false;
    V3 tvec = orig - v0;
    float u = dot(tvec, pvec);
    if (u < 0 || u > det) return false;
    if (past_u) *past_u = true;
    V3 qvec = cross(tvec, edge1);
    float v = dot(dir, qvec);
    if (v < 0 || u + v > det) return false;
    dist = dot(edge2, qvec) * (1.0f / det);
    return dist > kEps;
}

// glm::triangleNormal = normalize(cross(p1 - p2, p1 - p3)), include/glm/gtx/normal.inl:5-14, from
// the stored edges e1 = v1 - v0, e2 = v2 - v0.  v0 - v1 is -(v1 - v0) exactly, except that equal
// coordinates give +0 both ways round; 0.0f - e reproduces that (0 - (+0) = +0, 0 - x = -x), so
// the normal comes out with the reference's signed zeros too.
// Returns the cross product; the caller normalizes.
RTX_DEV V3 triNormalDir(V3 edge1, V3 edge2) {
    const V3 a = v3(0.0f - edge1.x, 0.0f - edge1.y, 0.0f - edge1.z);
    const V3 b = v3(0.0f - edge2.x, 0.0f - edge2.y, 0.0f - edge2.z);
    return cross(a, b);
}

// Cube::rayIntersect(orig, dir, nearDist, farDist), src/SceneObjects.cpp:101-139.  An origin
// inside the box returns true WITHOUT writing nearDist (the caller's 0 stays: "hit at distance 0").
RTX_DEV bool boxSlab(const float lo[3], const float hi[3], V3 orig, V3 dir, float &nearDist) {
    if (orig.x >= lo[0] && orig.x <= hi[0] && orig.y >= lo[1] && orig.y <= hi[1] && orig.z >= lo[2] && orig.z <= hi[2])
        return true;
    float t_near = kFltMin;  // +1.17549435e-38, as written at :108
    float t_far = kFltMax;
    const float o[3] = {orig.x, orig.y, orig.z};
    const float d[3] = {dir.x, dir.y, dir.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (fabsf(d[i]) >= kEps) {
            float t1 = (lo[i] - o[i]) / d[i];
            float t2 = (hi[i] - o[i]) / d[i];
            if (t1 > t2) {
                float t = t1;
                t1 = t2;
                t2 = t;
            }
            if (t1 > t_near) t_near = t1;
            if (t2 < t_far) t_far = t2;
            if (t_near > t_far) return false;
            if (t_far < kEps) return false;
        } else {
            if (o[i] < lo[i] || o[i] > hi[i]) return false;
        }
    }
    if (!(t_near <= t_far && t_far >= kEps)) return false;
    nearDist = t_near;
    return true;
}

// Conservative ray/box filter.  Uses the same (bound - origin) differences as the exact slab test
// but multiplies by an approximate reciprocal (clamped to +-1e30 so that a zero direction component
// gives a huge finite number, never inf * 0) and widens the far side by 2e-6, more than the
// rounding difference between this and the exact IEEE divisions.  It never rejects a ray that
// boxSlab accepts (an origin inside the box gives t_near <= 0 <= t_far here as well).
RTX_DEV bool slabMaybe(const float lo[3], const float hi[3], V3 o, V3 inv) {
    const float ax = (lo[0] - o.x) * inv.x, bx = (hi[0] - o.x) * inv.x;
    const float ay = (lo[1] - o.y) * inv.y, by = (hi[1] - o.y) * inv.y;
    const float az = (lo[2] - o.z) * inv.z, bz = (hi[2] - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return tn <= tf * 1.000002f;
}

RTX_DEV V3 safeInverse(V3 d) {
    return v3(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -1e30f, 1e30f),
              __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -1e30f, 1e30f),
              __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -1e30f, 1e30f));
}

// float -> int the way the reference's x86-64 build converts (cvttss2si): NaN and out-of-range
// give INT_MIN.  The plain device conversion saturates instead.
RTX_DEV int truncToInt(float v) {
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return (int)0x80000000;
    return (int)v;
}
RTX_DEV int truncToIntD(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
    return (int)v;
}

// Normal of Cube::rayIntersect(orig, dir, dist, n), src/SceneObjects.cpp:146-172, before its normalize (the caller's)
RTX_DEV V3 boxNormalDir(const float lo[3], const float hi[3], V3 hit) {
    V3 c = (v3(lo[0], lo[1], lo[2]) + v3(hi[0], hi[1], hi[2])) * 0.5f;
    V3 d = (v3(lo[0], lo[1], lo[2]) - v3(hi[0], hi[1], hi[2])) * 0.5f;
    V3 p = hit - c;
    const float bias = 1.000001f;
    float v1 = p.x / fabsf(d.x) * bias;
    float v2 = p.y / fabsf(d.y) * bias;
    float v3_ = p.z / fabsf(d.z) * bias;
    int iv1 = truncToInt(v1), iv2 = truncToInt(v2), iv3 = truncToInt(v3_);
    if (iv1 == 0 && iv2 == 0 && iv3 == 0) {
        float mx = v1 > v2 ? v1 : v2;
        mx = mx > v3_ ? mx : v3_;
        if (fabsf(v1 - mx) < kEps)
            iv1 = 1;
        else if (fabsf(v2 - mx) < kEps)
            iv2 = 1;
        else
            iv3 = 1;
    }
    return v3((float)iv1, (float)iv2, (float)iv3);
}

// the direction of a mirrored ray, src/Scene.cpp:97
RTX_DEV V3 reflectDir(V3 I, V3 N) { return normalize(reflect(I, N)); }

// glm::intersectRayPlane for the plane y = plane_y with normal (0,1,0)
// (include/glm/gtx/intersect.inl:6-26).  dot(dir, n) = dir.x*0 + dir.y*1 + dir.z*0 equals dir.y
// and dot(p0 - orig, n) equals plane_y - orig.y whenever the other components are finite (adding
// a signed zero changes nothing but the sign of a zero, which neither comparison below sees).
RTX_DEV bool planeHit(float plane_y, V3 orig, V3 dir, float &dist) {
    const float d = dir.y;
    const float num = plane_y - orig.y;
    // t = num / d is positive only if num and d have the same sign and num is not zero; deciding
    // that first skips the IEEE division for every ray that points away from the plane
    if (fabsf(d) > kEps && num != 0.0f && (__float_as_uint(num) ^ __float_as_uint(d)) < 0x80000000u) {
        const float t = num / d;
        if (t > 0.0f) {  // still needed: the quotient of tiny operands can underflow to 0
            dist = t;
            return true;
        }
    }
    return false;
}

// free refract(), src/Scene.cpp:19-27; the recursion there happens at most once (it flips N and
// swaps the indices when the ray leaves the medium).
RTX_DEV V3 refractDir(V3 I, V3 N, float etat) {
    float etai = 1.0f;
    float cosi = -fmaxf(-1.0f, fminf(1.0f, dot(I, N)));
    if (cosi < 0) {
        N = -N;
        float t = etai;
        etai = etat;
        etat = t;
        cosi = -fmaxf(-1.0f, fminf(1.0f, dot(I, N)));
        // a second flip cannot make progress; the reference would recurse again only for
        // cosi < 0 here, which needs dot(I,N) and dot(I,-N) both positive
    }
    float eta = etai / etat;
    float k = 1 - eta * eta * (1 - cosi * cosi);
    if (k < 0) return v3(1, 0, 0);
    return I * eta + N * (eta * cosi - sqrtf(k));
}

#ifdef RTX_FAST_MATH
#pragma clang fp contract(fast)  // the tolerance build: shading arithmetic (what follows this file) may fuse
#endif

}  // namespace rtx
