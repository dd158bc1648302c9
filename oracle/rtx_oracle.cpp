// TEST INFRASTRUCTURE ONLY -- the parity checker, never the product.
//
// CPU restatement of the reference's per-pixel trace loop behind the C ABI of include/rtx.h.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// It follows the reference operation by operation (same order of float operations, same strict
// comparisons, same quirks) so that, built with `-O3 -ffp-contract=off` and no -march flags, its
// frames are byte-identical to the reference binary's in the container where both were built.
// Pinned by: tests/test_oracle_golden.py (SHA-256 of whole frames rendered by the unmodified
// reference build, tests/golden/frames.json) and tests/test_oracle_vs_reference.py (call-by-call
// against oracle/_ref/libref.so when that has been built).
//
// Each function cites the reference lines it restates; paths are relative to /root/reference.
#include "rtx.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#include <omp.h>

namespace {

thread_local std::string g_err;
int fail(int code, const char *msg) {
    g_err = msg;
    return code;
}

// ---- the handful of GLM 0.9.9.7 formulas the reference uses, in GLM's operation order ----
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
// include/glm/detail/func_geometric.inl:48-55: tmp = a*b; tmp.x + tmp.y + tmp.z
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// func_geometric.inl:68-79
inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
// func_geometric.inl:82-90 with inversesqrt = 1/sqrt (func_exponential.inl:44-50)
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
// func_geometric.inl:104-110: I - N * dot(N, I) * 2
inline V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0f; }
// length(p1 - p0)
inline float distance(V3 p0, V3 p1) {
    V3 d = p1 - p0;
    return std::sqrt(dot(d, d));
}
// include/glm/gtx/normal.inl:5-14
inline V3 triangleNormal(V3 p1, V3 p2, V3 p3) { return normalize(cross(p1 - p2, p1 - p3)); }

// include/glm/gtx/intersect.inl:6-26
inline bool intersectRayPlane(V3 orig, V3 dir, V3 planeOrig, V3 planeNormal, float &dist) {
    float d = dot(dir, planeNormal);
    if (std::fabs(d) > FLT_EPSILON) {
        float t = dot(planeOrig - orig, planeNormal) / d;
        if (t > 0.0f) {
            dist = t;
            return true;
        }
    }
    return false;
}

struct Mat {  // include/Material.h:8-18
    float refractive;
    float albedo[4];
    V3 diffuse;
    float specular;
};

struct Box {  // Cube, include/SceneObjects.h:64-74
    V3 lo, hi;
    int mat;
};

struct Tri {
    V3 v0, v1, v2;
    int mat;
};

// Cube::load, src/SceneObjects.cpp:75-89
Box makeBox(V3 a, V3 b, int mat) {
    Box c;
    c.lo = v3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z));
    c.hi = v3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z));
    c.mat = mat;
    return c;
}

inline float comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// Cube::rayIntersect(orig, dir, nearDist, farDist), src/SceneObjects.cpp:101-139.
// An origin inside the box returns true WITHOUT writing nearDist/farDist.
bool boxSlab(const Box &c, V3 orig, V3 dir, float &nearDist, float &farDist) {
    if (orig.x >= c.lo.x && orig.x <= c.hi.x && orig.y >= c.lo.y && orig.y <= c.hi.y &&
        orig.z >= c.lo.z && orig.z <= c.hi.z)
        return true;
    float t_near = std::numeric_limits<float>::min();  // +1.17549435e-38, as written at :108
    float t_far = std::numeric_limits<float>::max();
    const float eps = FLT_EPSILON;
    for (int i = 0; i < 3; ++i) {
        float d = comp(dir, i), o = comp(orig, i), lo = comp(c.lo, i), hi = comp(c.hi, i);
        if (std::fabs(d) >= eps) {
            float t1 = (lo - o) / d;
            float t2 = (hi - o) / d;
            if (t1 > t2) std::swap(t1, t2);
            if (t1 > t_near) t_near = t1;
            if (t2 < t_far) t_far = t2;
            if (t_near > t_far) return false;
            if (t_far < eps) return false;
        } else {
            if (o < lo || o > hi) return false;
        }
    }
    if (!(t_near <= t_far && t_far >= eps)) return false;
    nearDist = t_near;
    farDist = t_far;
    return true;
}

// x86-64 float -> int conversion as cvttss2si performs it (what `int iv(v)` compiles to in the
// reference build): out-of-range and NaN give INT_MIN.
inline int truncToInt(float v) {
    if (!(v > -2147483904.0f && v < 2147483648.0f)) return std::numeric_limits<int>::min();
    return (int)v;
}

// double -> int as cvttsd2si performs it
inline int truncToIntD(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return std::numeric_limits<int>::min();
    return (int)v;
}

// Cube::rayIntersect(orig, dir, dist, n), src/SceneObjects.cpp:141-174
bool boxHit(const Box &c, V3 orig, V3 dir, float &dist, V3 &n) {
    float farDist = 0.0f;
    if (!boxSlab(c, orig, dir, dist, farDist)) {
        n = v3(0, 0, 0);
        return false;
    }
    V3 hit = orig + dir * dist;
    V3 ctr = (c.lo + c.hi) * 0.5f;
    V3 d = (c.lo - c.hi) * 0.5f;
    V3 p = hit - ctr;
    const float bias = 1.000001f;
    const float eps = FLT_EPSILON;
    float v1 = p.x / std::fabs(d.x) * bias;
    float v2 = p.y / std::fabs(d.y) * bias;
    float v3_ = p.z / std::fabs(d.z) * bias;
    int iv1 = truncToInt(v1), iv2 = truncToInt(v2), iv3 = truncToInt(v3_);
    if (iv1 == 0 && iv2 == 0 && iv3 == 0) {
        float mx = v1 > v2 ? v1 : v2;
        mx = mx > v3_ ? mx : v3_;
        if (std::fabs(v1 - mx) < eps)
            iv1 = 1;
        else if (std::fabs(v2 - mx) < eps)
            iv2 = 1;
        else
            iv3 = 1;
    }
    n = normalize(v3((float)iv1, (float)iv2, (float)iv3));
    return true;
}

// Sphere::rayIntersect, src/SceneObjects.cpp:15-26
inline bool sphereHit(V3 center, float radius, V3 orig, V3 dir, float &dist) {
    const float eps = FLT_EPSILON;
    V3 diff = center - orig;
    float t0 = dot(diff, dir);
    float dSquared = dot(diff, diff) - t0 * t0;
    float radius2 = radius * radius;
    if (dSquared > radius2) return false;
    float t1 = std::sqrt(radius2 - dSquared);
    dist = t0 > t1 + eps ? t0 - t1 : t0 + t1;
    return dist > eps;
}

// Triangle::rayIntersect, src/SceneObjects.cpp:28-49 (single-sided Moller-Trumbore)
inline bool triHit(const Tri &t, V3 orig, V3 dir, float &dist) {
    V3 edge1 = t.v1 - t.v0;
    V3 edge2 = t.v2 - t.v0;
    V3 pvec = cross(dir, edge2);
    float det = dot(edge1, pvec);
    const float eps = FLT_EPSILON;
    if (det < eps) return false;
    V3 tvec = orig - t.v0;
    float u = dot(tvec, pvec);
    if (u < 0 || u > det) return false;
    V3 qvec = cross(tvec, edge1);
    float v = dot(dir, qvec);
    if (v < 0 || u + v > det) return false;
    dist = dot(edge2, qvec) * (1.0f / det);
    return dist > eps;
}

// free refract(), src/Scene.cpp:19-27 (this overload, not glm::refract, is what the call at :99 binds)
V3 refractDir(V3 I, V3 N, float etat, float etai = 1.0f) {
    float cosi = -std::max(-1.0f, std::min(1.0f, dot(I, N)));
    if (cosi < 0) return refractDir(I, -N, etai, etat);
    float eta = etai / etat;
    float k = 1 - eta * eta * (1 - cosi * cosi);
    return k < 0 ? v3(1, 0, 0) : I * eta + N * (eta * cosi - sqrtf(k));
}

struct Island {
    float cx, cz, a, b;
};

struct Grid {  // BoundingBox, include/SceneObjects.h:76-99
    static const int size = 7;
    Box box;
    Box cell[size][size][size];
    std::vector<int> tris[size][size][size];
};

struct Stats {
    uint64_t primary = 0, secondary = 0, shadow = 0, sky = 0, box_hits = 0, cell_tests = 0, tri_tests = 0;
};

}  // namespace

struct rtx_ctx {
    std::vector<Mat> materials;
    std::vector<float> lights;  // xyzi
    std::vector<float> sx, sy, sz, sr;
    std::vector<int> smat;
    std::vector<Box> boxes;
    std::vector<Tri> tris;
    // mesh
    std::vector<Tri> mesh;
    bool has_mesh = false;
    Grid *grid = nullptr;
    // islands
    std::vector<Island> islands;
    float island_y = -4.0f;
    int island_mat = 0;
    V3 lightGreen{0, 0, 0}, darkGreen{0, 0, 0};
    // water
    std::vector<unsigned char> water;
    int ww = 0, wh = 0, water_mat = 0;
    float water_y = -4.5f, water_scale = 10.0f;
    // sky
    std::vector<unsigned char> sky;
    int skw = 0, skh = 0, sky_shift = 100;
    // options
    bool committed = false;
    bool count = false;
    bool prune = false;  // the oracle follows the reference and traces zero-weight children
    float far_clip = 700.0f;
    bool mesh_grid = true;
    int spp = 1;  // RTX_OPT_SPP
    float fprec = 0.001f;
    Stats stats;
    double last_ms = 0;

    ~rtx_ctx() { delete grid; }
};

namespace {

struct Hit {
    V3 point, N;
    Mat m;
};

// Sky::getColorAt, src/SceneObjects.cpp:440-445; texel = byte * (1/255.0f), :429-434
V3 skyColor(const rtx_ctx &s, V3 dir) {
    const int width = s.skw, height = s.skh;
    // atan2 is the double C function, acos the float one (overload resolution in the reference's
    // translation unit); both quotients are evaluated in double, then truncated to int.
    int a = std::max(0, std::min(width - 1, truncToIntD((atan2f(dir.z, dir.x) / (2 * M_PI) + .5) * width)));
    int b = std::max(0, std::min(height - 1, truncToIntD(acosf(dir.y) / M_PI * height)));
    unsigned placeu = (unsigned)a + ((unsigned)(b - s.sky_shift) % (unsigned)height) * (unsigned)width;
    int place = (int)placeu;
    size_t idx = (size_t)(long)place % ((size_t)width * height);
    const unsigned char *p = &s.sky[idx * 3];
    const float colorAlpha = 1 / 255.0f;
    return v3((float)p[0], (float)p[1], (float)p[2]) * colorAlpha;
}

// SceneObjects::modelsIntersect, src/SceneObjects.cpp:350-380
void meshIntersect(rtx_ctx &s, V3 orig, V3 dir, Hit &h, float &minDist, Stats *st) {
    float modelsDist = std::numeric_limits<float>::max();
    if (s.has_mesh) {
        const Grid &g = *s.grid;
        float nearDist = 0, farDist = 0;
        if (boxSlab(g.box, orig, dir, nearDist, farDist) && !s.mesh_grid) {
            // RTX_OPT_MESH_GRID = 0: every triangle is a candidate, in index order
            if (st) st->box_hits++;
            float curDist = 0.0f;
            for (const Tri &t : s.mesh) {
                if (st) st->tri_tests++;
                if (triHit(t, orig, dir, curDist) && curDist < modelsDist && curDist < minDist) {
                    modelsDist = curDist;
                    h.point = orig + dir * curDist;
                    h.N = triangleNormal(t.v0, t.v1, t.v2);
                    h.m = s.materials[t.mat];
                }
            }
        } else if (boxSlab(g.box, orig, dir, nearDist, farDist)) {
            if (st) st->box_hits++;
            for (int i = 0; i < Grid::size; i++)
                for (int j = 0; j < Grid::size; j++)
                    for (int k = 0; k < Grid::size; k++) {
                        const std::vector<int> &cellTris = g.tris[i][j][k];
                        if (cellTris.empty()) continue;
                        if (st) st->cell_tests++;
                        float cellFar = 0;
                        if (!boxSlab(g.cell[i][j][k], orig, dir, nearDist, cellFar)) continue;
                        float curDist = 0.0f;
                        for (int ti : cellTris) {
                            const Tri &t = s.mesh[ti];
                            if (st) st->tri_tests++;
                            if (triHit(t, orig, dir, curDist) && curDist < modelsDist && curDist < minDist) {
                                modelsDist = curDist;
                                h.point = orig + dir * curDist;
                                h.N = triangleNormal(t.v0, t.v1, t.v2);
                                h.m = s.materials[t.mat];
                            }
                        }
                    }
        }
    }
    minDist = std::min(modelsDist, minDist);
}

// Scene::sceneIntersect, src/Scene.cpp:29-58, with the five category loops of
// src/SceneObjects.cpp:289-317, 350-419 inlined in the reference's order.
bool sceneIntersect(rtx_ctx &s, V3 orig, V3 dir, Hit &h, Stats *st) {
    const float FMAX = std::numeric_limits<float>::max();
    float minDist = FMAX;
    {  // spheresIntersect :289-302
        float spheresDist = FMAX;
        for (size_t i = 0; i < s.sx.size(); ++i) {
            float curDist = 0.0f;
            V3 c = v3(s.sx[i], s.sy[i], s.sz[i]);
            if (sphereHit(c, s.sr[i], orig, dir, curDist) && curDist < spheresDist && curDist < minDist) {
                spheresDist = curDist;
                h.point = orig + dir * curDist;
                h.N = normalize(h.point - c);
                h.m = s.materials[s.smat[i]];
            }
        }
        minDist = std::min(spheresDist, minDist);
    }
    {  // trianglesIntersect :304-317
        float trianglesDist = FMAX;
        for (const Tri &t : s.tris) {
            float curDist = 0.0f;
            if (triHit(t, orig, dir, curDist) && curDist < trianglesDist && curDist < minDist) {
                trianglesDist = curDist;
                h.point = orig + dir * curDist;
                h.N = triangleNormal(t.v0, t.v1, t.v2);
                h.m = s.materials[t.mat];
            }
        }
        minDist = std::min(trianglesDist, minDist);
    }
    {  // cubesIntersect :382-397
        float cubesDist = FMAX;
        for (const Box &c : s.boxes) {
            float curDist = 0.0f;
            V3 normal = v3(0, 0, 0);
            if (boxHit(c, orig, dir, curDist, normal) && curDist < cubesDist && curDist < minDist) {
                cubesDist = curDist;
                h.point = orig + dir * curDist;
                h.N = normal;
                h.m = s.materials[c.mat];
            }
        }
        minDist = std::min(cubesDist, minDist);
    }
    meshIntersect(s, orig, dir, h, minDist, st);

    // water plane, src/Scene.cpp:38-54
    float waterDist = FMAX;
    if (s.ww > 0) {
        V3 waterOrig = v3(0.0f, s.water_y, 0.0f);
        V3 waterNorm = v3(0.0f, 1.0f, 0.0f);
        if (intersectRayPlane(orig, dir, waterOrig, waterNorm, waterDist) && waterDist < minDist) {
            minDist = waterDist;
            h.point = orig + dir * waterDist;
            h.N = waterNorm;
            h.m = s.materials[s.water_mat];
            float scale = s.water_scale;
            float x = h.point.x / scale, z = h.point.z / scale;
            float xx = x - std::floor(x), zz = z - std::floor(z);
            unsigned xpos = (unsigned)(xx * s.ww), zpos = (unsigned)(zz * s.wh);
            // xx can round up to exactly 1.0f for tiny negative x; the reference then reads one
            // element past the row (past the vector for the last row).  Clamp to the last texel.
            size_t idx = std::min((size_t)zpos * s.ww + xpos, (size_t)s.ww * s.wh - 1);
            const unsigned char *p = &s.water[idx * 3];
            // texel = colour * 1.0f / 255.0f (a true divide), src/Scene.cpp:246
            V3 tex = v3((float)p[0], (float)p[1], (float)p[2]) * 1.0f / 255.0f;
            h.m.diffuse = (tex * 0.3f + h.m.diffuse) / 2.0f;
        }
        minDist = std::min(waterDist, minDist);
    }

    {  // islandsIntersect :399-419 with Island::rayIntersect :273-287
        float islandsDist = FMAX;
        for (const Island &is : s.islands) {
            float curDist = 0.0f;
            bool ok = false;
            {
                float dist = curDist;
                if (intersectRayPlane(orig, dir, v3(0.0f, s.island_y, 0.0f), v3(0.0f, 1.0f, 0.0f), dist)) {
                    V3 pt = orig + dir * dist;
                    float xa = (pt.x - is.cx) / is.a, zb = (pt.z - is.cz) / is.b;
                    if (pt.x > is.cx - is.a && pt.x < is.cx + is.a && pt.z > is.cz - is.b &&
                        pt.z < is.cz + is.b && xa * xa + zb * zb <= 1.0f) {
                        ok = true;
                        curDist = dist;
                    }
                }
            }
            if (ok && curDist < minDist) {
                islandsDist = curDist;
                h.point = orig + dir * curDist;
                h.N = v3(0.0f, 1.0f, 0.0f);
                h.m = s.materials[s.island_mat];
                int parity = truncToInt(h.point.x + h.point.z) & 1;
                h.m.diffuse = parity ? s.lightGreen : s.darkGreen;
                float x = h.point.x, fx = roundf(h.point.x);
                float z = (float)((double)(ceilf(h.point.x + h.point.z) - fx) - 0.5);
                float rad = 0.15f;
                if ((x - fx) * (x - fx) + (h.point.z - z) * (h.point.z - z) < rad * rad)
                    h.m.diffuse = parity ? s.darkGreen : s.lightGreen;
            }
        }
        minDist = std::min(islandsDist, minDist);
    }
    return minDist < s.far_clip;
}

struct Lightning {
    float diffuse = 0, specular = 0;
};

// Scene::calcLightning, src/Scene.cpp:60-81
Lightning calcLightning(rtx_ctx &s, V3 dir, V3 point, V3 N, const Mat &mt, Stats *st) {
    Lightning l;
    V3 positiveOrig = point + N * s.fprec, negativeOrig = point - N * s.fprec;
    for (size_t i = 0; i + 3 < s.lights.size(); i += 4) {
        V3 lpos = v3(s.lights[i], s.lights[i + 1], s.lights[i + 2]);
        float intensity = s.lights[i + 3];
        V3 lightDir = normalize(lpos - point);
        if (s.prune) {
            // RTX_OPT_PRUNE: a light that can add to neither sum needs no shadow ray (same rule, same
            // ray counts as the HIP backend; the sums are unchanged)
            bool lightsPlain = true;
            for (size_t k = 3; k < s.lights.size(); k += 4) lightsPlain = lightsPlain && s.lights[k] >= 0.0f && s.lights[k] <= 3.0e38f;
            const bool diffuseCounts = dot(lightDir, N) >= 0 && !(mt.albedo[0] == 0.0f && lightsPlain);
            if (!diffuseCounts) {
                const float sd = dot(-reflect(-lightDir, N), dir);
                bool specCounts = sd >= 0;
                if (specCounts && lightsPlain) specCounts = !(powf(sd, mt.specular) * intensity == 0.0f);
                if (!specCounts) continue;
            }
        }
        V3 shadowOrig = std::signbit(dot(lightDir, N)) ? negativeOrig : positiveOrig;
        Hit sh;
        sh.point = v3(0, 0, 0);
        sh.N = v3(0, 0, 0);
        if (st) st->shadow++;
        if (sceneIntersect(s, shadowOrig, lightDir, sh, st) &&
            distance(sh.point, shadowOrig) < distance(lpos, point))
            continue;
        float difIntDot = dot(lightDir, N);
        if (difIntDot >= 0) l.diffuse += intensity * difIntDot;
        float specIntDot = dot(-reflect(-lightDir, N), dir);
        if (specIntDot >= 0) l.specular += powf(specIntDot, mt.specular) * intensity;
    }
    return l;
}

// Scene::traceRay, src/Scene.cpp:83-106
V3 traceRay(rtx_ctx &s, V3 orig, V3 dir, int depth, int fullDepth, Stats *st) {
    Hit h;
    h.point = v3(0, 0, 0);
    h.N = v3(0, 0, 0);
    bool hit = false;
    if (depth > 0) {
        if (st) (depth == fullDepth ? st->primary : st->secondary)++;
        hit = sceneIntersect(s, orig, dir, h, st);
    }
    if (!hit) {
        if (st) st->sky++;
        return skyColor(s, dir);
    }
    Lightning light = calcLightning(s, dir, h.point, h.N, h.m, st);
    V3 positiveOrig = h.point + h.N * s.fprec, negativeOrig = h.point - h.N * s.fprec;

    V3 reflectColor = v3(0, 0, 0), refractColor = v3(0, 0, 0);
    const bool doRefl = !s.prune || h.m.albedo[2] != 0.0f;
    const bool doRefr = !s.prune || h.m.albedo[3] != 0.0f;
    if (doRefl) {
        V3 reflectDir = normalize(reflect(dir, h.N));
        V3 reflectOrig = std::signbit(dot(reflectDir, h.N)) ? negativeOrig : positiveOrig;
        reflectColor = traceRay(s, reflectOrig, reflectDir, depth - 1, fullDepth, st);
    }
    if (doRefr) {
        V3 rd = normalize(refractDir(dir, h.N, h.m.refractive));
        V3 refractOrig = std::signbit(dot(rd, h.N)) ? negativeOrig : positiveOrig;
        refractColor = traceRay(s, refractOrig, rd, depth - 1, fullDepth, st);
    }
    V3 c = h.m.diffuse * light.diffuse * h.m.albedo[0] + v3(1.0f, 1.0f, 1.0f) * light.specular * h.m.albedo[1];
    if (doRefl) c = c + reflectColor * h.m.albedo[2];
    if (doRefr) c = c + refractColor * h.m.albedo[3];
    return c;
}

// BoundingBox::init + initTriangles, src/SceneObjects.cpp:176-209
void buildGrid(rtx_ctx &s, V3 lo, V3 hi) {
    delete s.grid;
    Grid *g = new Grid();
    g->box = makeBox(lo, hi, 0);
    const int size = Grid::size;
    float dx0 = g->box.lo.x, dx1 = g->box.hi.x, dy0 = g->box.lo.y, dy1 = g->box.hi.y, dz0 = g->box.lo.z,
          dz1 = g->box.hi.z;
    float lx = dx1 - dx0, ly = dy1 - dy0, lz = dz1 - dz0;
    float px = lx / float(size), py = ly / float(size), pz = lz / float(size);
    for (int i = 0; i < size; i++)
        for (int j = 0; j < size; j++)
            for (int k = 0; k < size; k++) {
                V3 a = v3(dx0 + px * i, dy0 + py * j, dz0 + pz * k);
                V3 b = v3(dx0 + px * (i + 1), dy0 + py * (j + 1), dz0 + pz * (k + 1));
                g->cell[i][j][k] = makeBox(a, b, 0);
            }
    for (size_t ti = 0; ti < s.mesh.size(); ++ti) {
        const Tri &t = s.mesh[ti];
        const V3 vs[3] = {t.v0, t.v1, t.v2};
        for (const V3 &v : vs) {
            float fx = (v.x - dx0) * size / lx;
            float fy = (v.y - dy0) * size / ly;
            float fz = (v.z - dz0) * size / lz;
            int i = int(fx), j = int(fy), k = int(fz);
            if (i < 0 || j < 0 || k < 0 || i >= size || j >= size || k >= size) continue;  // reference: UB
            std::vector<int> &cell = g->tris[i][j][k];
            // the reference looks the whole list through for ti; ti can only be its last element (a
            // triangle's three vertices are entered one after the other)
            if (cell.empty() || cell.back() != (int)ti) cell.push_back((int)ti);
        }
    }
    s.grid = g;
}

int checkMat(const rtx_ctx *c, const int *mat, int n) {
    for (int i = 0; i < n; ++i)
        if (mat[i] < 0 || (size_t)mat[i] >= c->materials.size()) return fail(RTX_E_INVALID, "material index out of range");
    return RTX_OK;
}

}  // namespace

extern "C" {

const char *rtx_last_error(void) { return g_err.c_str(); }

int rtx_create(rtx_ctx **out, int /*device_id*/) {
    if (!out) return fail(RTX_E_INVALID, "rtx_create: out is null");
    *out = new rtx_ctx();
    return RTX_OK;
}

void rtx_destroy(rtx_ctx *ctx) { delete ctx; }

int rtx_set_option(rtx_ctx *c, int option, uint32_t value) {
    if (!c) return fail(RTX_E_INVALID, "null context");
    float f;
    memcpy(&f, &value, 4);
    switch (option) {
        case RTX_OPT_STATS: c->count = value != 0; break;
        case RTX_OPT_PRUNE: c->prune = value != 0; break;
        case RTX_OPT_FAR_CLIP: c->far_clip = f; break;
        case RTX_OPT_SURFACE_BIAS: c->fprec = f; break;
        case RTX_OPT_MESH_GRID: c->mesh_grid = value != 0; break;
        case RTX_OPT_BVH_BUILDER: break;  // no tree here
        case RTX_OPT_SCHEDULE_FEEDBACK: break;  // no tile schedule here
        case RTX_OPT_QUAD: break;
        case RTX_OPT_PIPELINE: break;  // one loop here
        case RTX_OPT_PIPELINE_BUDGET_MB: break;
        case RTX_OPT_PIN_OUTPUT: break;  // no device to copy from
        case RTX_OPT_TILE_PROBE: break;  // no tile schedule here
        case RTX_OPT_QUAD_KERNEL: break;
        case RTX_OPT_RAY_POOL: break;  // no waves here
        case RTX_OPT_SPP:
            if (value != 1 && value != 4) return fail(RTX_E_INVALID, "RTX_OPT_SPP: 1 or 4");
            c->spp = (int)value;
            break;
        default: return fail(RTX_E_INVALID, "unknown option");
    }
    return RTX_OK;
}

#define NEED(c)                                                    \
    if (!(c)) return fail(RTX_E_INVALID, "null context");          \
    if ((c)->committed) return fail(RTX_E_STATE, "scene is frozen after rtx_commit")

int rtx_set_materials(rtx_ctx *c, const float *m9, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && !m9)) return fail(RTX_E_INVALID, "rtx_set_materials: bad arguments");
    c->materials.resize(n);
    for (int i = 0; i < n; ++i) {
        const float *m = m9 + 9 * i;
        Mat &o = c->materials[i];
        o.refractive = m[0];
        for (int k = 0; k < 4; ++k) o.albedo[k] = m[1 + k];
        o.diffuse = v3(m[5], m[6], m[7]);
        o.specular = m[8];
    }
    return RTX_OK;
}

int rtx_set_lights(rtx_ctx *c, const float *xyzi, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && !xyzi)) return fail(RTX_E_INVALID, "rtx_set_lights: bad arguments");
    c->lights.assign(xyzi, xyzi + 4 * (size_t)n);
    return RTX_OK;
}

int rtx_set_spheres(rtx_ctx *c, const float *cx, const float *cy, const float *cz, const float *r,
                    const int *mat, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && (!cx || !cy || !cz || !r || !mat))) return fail(RTX_E_INVALID, "rtx_set_spheres: bad arguments");
    if (int e = checkMat(c, mat, n)) return e;
    c->sx.assign(cx, cx + n), c->sy.assign(cy, cy + n), c->sz.assign(cz, cz + n), c->sr.assign(r, r + n);
    c->smat.assign(mat, mat + n);
    return RTX_OK;
}

int rtx_set_boxes(rtx_ctx *c, const float *a, const float *b, const int *mat, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && (!a || !b || !mat))) return fail(RTX_E_INVALID, "rtx_set_boxes: bad arguments");
    if (int e = checkMat(c, mat, n)) return e;
    c->boxes.clear();
    for (int i = 0; i < n; ++i)
        c->boxes.push_back(makeBox(v3(a[3 * i], a[3 * i + 1], a[3 * i + 2]), v3(b[3 * i], b[3 * i + 1], b[3 * i + 2]), mat[i]));
    return RTX_OK;
}

int rtx_set_triangles(rtx_ctx *c, const float *v9, const int *mat, int n) {
    NEED(c);
    if (n < 0 || (n > 0 && (!v9 || !mat))) return fail(RTX_E_INVALID, "rtx_set_triangles: bad arguments");
    if (int e = checkMat(c, mat, n)) return e;
    c->tris.clear();
    for (int i = 0; i < n; ++i) {
        const float *v = v9 + 9 * i;
        c->tris.push_back(Tri{v3(v[0], v[1], v[2]), v3(v[3], v[4], v[5]), v3(v[6], v[7], v[8]), mat[i]});
    }
    return RTX_OK;
}

int rtx_set_mesh(rtx_ctx *c, const float *verts, int nv, const int *idx, int nt, int mat) {
    NEED(c);
    if (nv < 0 || nt < 0 || (nv > 0 && !verts) || (nt > 0 && !idx)) return fail(RTX_E_INVALID, "rtx_set_mesh: bad arguments");
    if (nt > 0)
        if (int e = checkMat(c, &mat, 1)) return e;
    c->mesh.clear();
    c->has_mesh = false;
    delete c->grid;
    c->grid = nullptr;
    if (nv == 0) return RTX_OK;
    // bounding box over all vertices, src/SceneObjects.cpp:222-243
    float min_x, min_y, min_z, max_x, max_y, max_z;
    min_x = min_y = min_z = 1000;
    max_x = max_y = max_z = -1000;
    for (int i = 0; i < nv; ++i) {
        const float *v = verts + 3 * i;
        min_x = std::min(min_x, v[0]), min_y = std::min(min_y, v[1]), min_z = std::min(min_z, v[2]);
        max_x = std::max(max_x, v[0]), max_y = std::max(max_y, v[1]), max_z = std::max(max_z, v[2]);
    }
    for (int i = 0; i < nt; ++i) {
        const int *f = idx + 3 * i;
        for (int k = 0; k < 3; ++k)
            if (f[k] < 0 || f[k] >= nv) return fail(RTX_E_INVALID, "rtx_set_mesh: vertex index out of range");
        auto V = [&](int j) { return v3(verts[3 * j], verts[3 * j + 1], verts[3 * j + 2]); };
        c->mesh.push_back(Tri{V(f[0]), V(f[1]), V(f[2]), mat});
    }
    float eps = 0.0001f;
    V3 veps = v3(eps, eps, eps);
    buildGrid(*c, v3(min_x, min_y, min_z) - veps, v3(max_x, max_y, max_z) + veps);
    c->has_mesh = true;
    return RTX_OK;
}

int rtx_set_islands(rtx_ctx *c, const float *p, int n, float plane_y, int mat, const float *light_rgb,
                    const float *dark_rgb) {
    NEED(c);
    if (n < 0 || (n > 0 && (!p || !light_rgb || !dark_rgb))) return fail(RTX_E_INVALID, "rtx_set_islands: bad arguments");
    if (n > 0)
        if (int e = checkMat(c, &mat, 1)) return e;
    c->islands.clear();
    for (int i = 0; i < n; ++i) c->islands.push_back(Island{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]});
    c->island_y = plane_y;
    c->island_mat = mat;
    if (n > 0) {
        c->lightGreen = v3(light_rgb[0], light_rgb[1], light_rgb[2]);
        c->darkGreen = v3(dark_rgb[0], dark_rgb[1], dark_rgb[2]);
    }
    return RTX_OK;
}

int rtx_set_water(rtx_ctx *c, const unsigned char *rgb, int w, int h, float plane_y, float scale, int mat) {
    NEED(c);
    if (w < 0 || h < 0 || ((w > 0 || h > 0) && (!rgb || w == 0 || h == 0))) return fail(RTX_E_INVALID, "rtx_set_water: bad arguments");
    if (w > 0)
        if (int e = checkMat(c, &mat, 1)) return e;
    c->ww = w, c->wh = h, c->water_y = plane_y, c->water_scale = scale, c->water_mat = mat;
    c->water.assign(rgb, rgb + (size_t)w * h * 3);
    return RTX_OK;
}

int rtx_set_envmap(rtx_ctx *c, const unsigned char *rgb, int w, int h, int row_shift) {
    NEED(c);
    if (w <= 0 || h <= 0 || !rgb) return fail(RTX_E_INVALID, "rtx_set_envmap: bad arguments");
    c->skw = w, c->skh = h, c->sky_shift = row_shift;
    c->sky.assign(rgb, rgb + (size_t)w * h * 3);
    return RTX_OK;
}

int rtx_commit(rtx_ctx *c) {
    if (!c) return fail(RTX_E_INVALID, "null context");
    if (c->committed) return fail(RTX_E_STATE, "already committed");
    if (c->skw == 0) return fail(RTX_E_STATE, "rtx_commit: no environment map (the reference cannot render without one)");
    if (c->materials.empty()) return fail(RTX_E_STATE, "rtx_commit: no materials");
    c->committed = true;
    return RTX_OK;
}

int rtx_frame_rows(const rtx_frame *f) {
    if (!f || f->row_step <= 0 || f->row_first < 0 || f->row_first >= f->height) return 0;
    return (f->height - f->row_first + f->row_step - 1) / f->row_step;
}

static int checkFrame(const rtx_ctx *c, const rtx_camera *cam, const rtx_frame *f) {
    if (!c || !cam || !f) return fail(RTX_E_INVALID, "null argument");
    if (!c->committed) return fail(RTX_E_STATE, "render before rtx_commit");
    if (f->width <= 0 || f->height <= 0 || f->row_step <= 0 || f->row_first < 0)
        return fail(RTX_E_INVALID, "bad frame description");
    if (f->depth < 0 || f->depth > RTX_MAX_DEPTH) return fail(RTX_E_LIMIT, "depth outside [0, RTX_MAX_DEPTH]");
    return RTX_OK;
}

// primary ray direction of pixel (row i, column j), src/Scene.cpp:217-220
// (dx, dy: offset of the sample from the pixel centre, 0 for the reference's one sample per pixel)
static inline V3 primaryDir(const rtx_camera *cam, int width, int height, size_t i, size_t j, float dx = 0.0f, float dy = 0.0f) {
    const float *M = cam->basis;
    float x = ((j + 0.5f) + dx) - width / 2.0f;
    float y = -((i + 0.5f) + dy) + height / 2.0f;
    float z = -height / (2.0f * cam->tan_half_fov);
    // mat3 * vec3, include/glm/detail/type_mat3x3.inl:468-474
    V3 d = v3(M[0] * x + M[3] * y + M[6] * z, M[1] * x + M[4] * y + M[7] * z, M[2] * x + M[5] * y + M[8] * z);
    if (cam->add_origin) d = d + v3(cam->origin[0], cam->origin[1], cam->origin[2]);
    return normalize(d);
}

// oracle-only: the primary directions of a frame, rows x width x 3 floats
int orx_primary_dirs(const rtx_camera *cam, const rtx_frame *f, float *out) {
    const int rows = rtx_frame_rows(f);
    for (int k = 0; k < rows; ++k)
        for (int j = 0; j < f->width; ++j) {
            V3 d = primaryDir(cam, f->width, f->height, (size_t)f->row_first + (size_t)k * f->row_step, (size_t)j);
            float *o = out + ((size_t)k * f->width + j) * 3;
            o[0] = d.x, o[1] = d.y, o[2] = d.z;
        }
    return RTX_OK;
}

// pixel loop of Scene::render, src/Scene.cpp:214-224, with getRGBColor :204-210
int rtx_render(rtx_ctx *c, const rtx_camera *cam, const rtx_frame *f, unsigned char *bgr_out, rtx_stats *stats) {
    if (int e = checkFrame(c, cam, f)) return e;
    const int rows = rtx_frame_rows(f);
    if (rows > 0 && !bgr_out) return fail(RTX_E_INVALID, "null output buffer");
    const int width = f->width, height = f->height;
    const V3 camera = v3(cam->origin[0], cam->origin[1], cam->origin[2]);
    Stats total;
    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel
    {
        Stats local;
        Stats *st = c->count ? &local : nullptr;
        // (pixels are independent: rows and columns are dealt out together, so that a frame of a few
        // long rows still uses every thread)
#pragma omp for collapse(2) schedule(dynamic, 64)
        for (int k = 0; k < rows; ++k) {
            for (size_t j = 0; j < (size_t)width; ++j) {
                const size_t i = (size_t)f->row_first + (size_t)k * f->row_step;
                unsigned char *row = bgr_out + (size_t)k * width * 3;
                V3 col;
                if (c->spp == 1) {
                    V3 dir = primaryDir(cam, width, height, i, j);
                    col = traceRay(*c, camera, dir, f->depth, f->depth, st);
                } else {  // RTX_OPT_SPP = 4: rotated grid, summed in sample order, times 0.25
                    static const float dx[4] = {0.125f, -0.375f, 0.375f, -0.125f}, dy[4] = {0.375f, 0.125f, -0.125f, -0.375f};
                    col = traceRay(*c, camera, primaryDir(cam, width, height, i, j, dx[0], dy[0]), f->depth, f->depth, st);
                    for (int k = 1; k < 4; ++k)
                        col = col + traceRay(*c, camera, primaryDir(cam, width, height, i, j, dx[k], dy[k]), f->depth, f->depth, st);
                    col = col * 0.25f;
                }
                unsigned char red = (unsigned char)(255 * std::max(0.0f, std::min(1.0f, col.x)));
                unsigned char green = (unsigned char)(255 * std::max(0.0f, std::min(1.0f, col.y)));
                unsigned char blue = (unsigned char)(255 * std::max(0.0f, std::min(1.0f, col.z)));
                row[3 * j + 0] = blue, row[3 * j + 1] = green, row[3 * j + 2] = red;  // bitmap_image.hpp:207-209
            }
        }
#pragma omp critical
        {
            total.primary += local.primary, total.secondary += local.secondary, total.shadow += local.shadow;
            total.sky += local.sky, total.box_hits += local.box_hits, total.cell_tests += local.cell_tests;
            total.tri_tests += local.tri_tests;
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    c->stats = total;
    c->last_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (stats) rtx_get_stats(c, stats);
    return RTX_OK;
}

int rtx_render_device(rtx_ctx *, const rtx_camera *, const rtx_frame *, void *, void *) {
    return fail(RTX_E_STATE, "the CPU oracle has no device path");
}

int rtx_render_device_noalloc(rtx_ctx *, const rtx_camera *, const rtx_frame *, void *, void *) {
    return fail(RTX_E_STATE, "the CPU oracle has no device path");
}

// Groups: the oracle has no devices, so a "device" is just another context; rank r renders rows r, r + n, ...
// with the same loop and the rows are put back in place on the host.  (Checks the partition arithmetic of
// the product's rtx_group_render and keeps the host layer's command line usable on this build.)
struct rtx_group {
    std::vector<rtx_ctx *> ctx;
};
int rtx_group_create(rtx_group **out, const int *device_ids, int n_devices) {
    if (!out || !device_ids || n_devices <= 0 || n_devices > 64) return fail(RTX_E_INVALID, "rtx_group_create: 1 to 64 ranks");
    rtx_group *g = new rtx_group();
    for (int r = 0; r < n_devices; ++r) {
        rtx_ctx *c = nullptr;
        if (int e = rtx_create(&c, device_ids[r])) {
            rtx_group_destroy(g);
            return e;
        }
        g->ctx.push_back(c);
    }
    *out = g;
    return RTX_OK;
}
void rtx_group_destroy(rtx_group *g) {
    if (!g) return;
    for (rtx_ctx *c : g->ctx) rtx_destroy(c);
    delete g;
}
int rtx_group_size(const rtx_group *g) { return g ? (int)g->ctx.size() : 0; }
rtx_ctx *rtx_group_ctx(rtx_group *g, int rank) { return g && rank >= 0 && rank < (int)g->ctx.size() ? g->ctx[rank] : nullptr; }
int rtx_group_commit(rtx_group *g) {
    if (!g) return fail(RTX_E_INVALID, "null group");
    for (rtx_ctx *c : g->ctx)
        if (int e = rtx_commit(c)) return e;
    return RTX_OK;
}
int rtx_group_render(rtx_group *g, const rtx_camera *cam, int width, int height, int depth, unsigned char *bgr_out, rtx_stats *stats) {
    if (!g || !cam || !bgr_out || width <= 0 || height <= 0) return fail(RTX_E_INVALID, "rtx_group_render: bad arguments");
    const int G = (int)g->ctx.size();
    const size_t row_bytes = (size_t)width * 3;
    if (stats) memset(stats, 0, sizeof *stats);
    for (int r = 0; r < G; ++r) {
        const rtx_frame fr = {width, height, depth, r, G};
        const int rows = rtx_frame_rows(&fr);
        std::vector<unsigned char> mine((size_t)rows * row_bytes);
        rtx_stats s;
        if (int e = rtx_render(g->ctx[r], cam, &fr, mine.data(), &s)) return e;
        for (int k = 0; k < rows; ++k) memcpy(bgr_out + (size_t)(r + k * G) * row_bytes, mine.data() + (size_t)k * row_bytes, row_bytes);
        if (stats) {
            uint64_t *acc = reinterpret_cast<uint64_t *>(stats);
            const uint64_t *add = reinterpret_cast<const uint64_t *>(&s);
            for (size_t k = 0; k < offsetof(rtx_stats, kernel_ms) / sizeof(uint64_t); ++k) acc[k] += add[k];
            stats->kernel_ms += s.kernel_ms;
        }
    }
    return RTX_OK;
}
const char *rtx_group_last_error(void) { return rtx_last_error(); }
int rtx_group_set_option(rtx_group *g, int option, uint32_t value) {
    if (!g) return fail(RTX_E_INVALID, "null group");
    for (rtx_ctx *c : g->ctx)
        if (int e = rtx_set_option(c, option, value)) return e;
    return RTX_OK;
}
int rtx_group_render_device(rtx_group *, const rtx_camera *, int, int, int, void **) { return fail(RTX_E_STATE, "the CPU oracle has no device memory"); }
void *rtx_group_root_stream(rtx_group *) { return nullptr; }
int rtx_group_wait(rtx_group *g) { return g ? RTX_OK : fail(RTX_E_INVALID, "null group"); }
void *rtx_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void rtx_host_free(void *p) { free(p); }

int rtx_deinterleave_rows_device(rtx_ctx *, const void *, int, int, int, int, void *, void *) {
    return fail(RTX_E_STATE, "the CPU oracle has no device path");
}

int rtx_get_stats(rtx_ctx *c, rtx_stats *o) {
    if (!c || !o) return fail(RTX_E_INVALID, "null argument");
    memset(o, 0, sizeof *o);
    o->primary_rays = c->stats.primary, o->secondary_rays = c->stats.secondary, o->shadow_rays = c->stats.shadow;
    o->sky_lookups = c->stats.sky, o->mesh_box_hits = c->stats.box_hits;
    o->bvh_node_tests = c->stats.cell_tests;  // for the oracle: non-empty grid cells tested
    o->triangle_tests = c->stats.tri_tests;
    o->kernel_ms = c->last_ms;
    return RTX_OK;
}

int rtx_trace_rays(rtx_ctx *c, const float *orig, const float *dir, int n, int depth, float *rgb_out) {
    if (!c || n < 0 || (n > 0 && (!orig || !dir || !rgb_out))) return fail(RTX_E_INVALID, "rtx_trace_rays: bad arguments");
    if (!c->committed) return fail(RTX_E_STATE, "trace before rtx_commit");
    if (depth < 0 || depth > RTX_MAX_DEPTH) return fail(RTX_E_LIMIT, "depth outside [0, RTX_MAX_DEPTH]");
    // (ray counts and the time of the loop are kept like rtx_render's: bench.py times a sample of rays this way)
    Stats total;
    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel
    {
        Stats local;
        Stats *st = c->count ? &local : nullptr;
#pragma omp for schedule(dynamic, 16)
        for (int i = 0; i < n; ++i) {
            V3 col = traceRay(*c, v3(orig[3 * i], orig[3 * i + 1], orig[3 * i + 2]),
                              v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), depth, depth, st);
            rgb_out[3 * i] = col.x, rgb_out[3 * i + 1] = col.y, rgb_out[3 * i + 2] = col.z;
        }
#pragma omp critical
        {
            total.primary += local.primary, total.secondary += local.secondary, total.shadow += local.shadow;
            total.sky += local.sky, total.box_hits += local.box_hits, total.cell_tests += local.cell_tests;
            total.tri_tests += local.tri_tests;
        }
    }
    c->stats = total;
    c->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RTX_OK;
}

int rtx_intersect_rays(rtx_ctx *c, const float *orig, const float *dir, int n, float *out16) {
    if (!c || n < 0 || (n > 0 && (!orig || !dir || !out16))) return fail(RTX_E_INVALID, "rtx_intersect_rays: bad arguments");
    if (!c->committed) return fail(RTX_E_STATE, "intersect before rtx_commit");
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; ++i) {
        Hit h;
        h.point = v3(0, 0, 0), h.N = v3(0, 0, 0);
        h.m = Mat{1.0f, {1, 0, 0, 0}, v3(0, 0, 0), 0.0f};  // Material(), include/Material.h:14
        bool ok = sceneIntersect(*c, v3(orig[3 * i], orig[3 * i + 1], orig[3 * i + 2]),
                                 v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), h, nullptr);
        float *o = out16 + 16 * (size_t)i;
        o[0] = ok ? 1.0f : 0.0f;
        o[1] = h.point.x, o[2] = h.point.y, o[3] = h.point.z;
        o[4] = h.N.x, o[5] = h.N.y, o[6] = h.N.z;
        o[7] = h.m.refractive;
        for (int k = 0; k < 4; ++k) o[8 + k] = h.m.albedo[k];
        o[12] = h.m.diffuse.x, o[13] = h.m.diffuse.y, o[14] = h.m.diffuse.z;
        o[15] = h.m.specular;
    }
    return RTX_OK;
}

// ---------------------------------------------------------------------------------------------
// Oracle-only helpers (not part of rtx.h): an independent restatement of the reference's scene
// constants and file formats, so the product's host layer can be checked against it too.
// ---------------------------------------------------------------------------------------------

// Materials table, src/Material.cpp:7-17; row = refractive, albedo[4], diffuse[3], specular.
// Constructor arguments are double literals narrowed to float (include/Material.h:15-17).
int orx_default_materials(float *m9 /*9 x 9*/) {
    auto put = [&](int i, double r, double a0, double a1, double a2, double a3, V3 d, double sp) {
        float *m = m9 + 9 * i;
        m[0] = (float)r, m[1] = (float)a0, m[2] = (float)a1, m[3] = (float)a2, m[4] = (float)a3;
        m[5] = d.x, m[6] = d.y, m[7] = d.z, m[8] = (float)sp;
    };
    put(0, 1, 1, 0, 0, 0, v3(0, 0, 0), 0);  // Material(): include/Material.h:14
    put(1, 1.5, 0.0, 0.5, 0.1, 0.8, v3((float)0.6, (float)0.7, (float)0.8), 125.);           // glass
    put(2, 1.0, 0.0, 10.0, 0.8, 0.0, v3(1.0f, 1.0f, 1.0f), 1425.);                            // mirror
    put(3, 1.0, 0.5, 3.0, 0.2, 0.0, v3(120 / 255.0f, 85 / 255.0f, 15 / 255.0f), 125);        // gold
    put(4, 1.0, 0.5, 3.0, 0.2, 0.0, v3((float)0.3, (float)0.3, (float)0.3), 125);            // silver
    put(5, 1.0, 0.9, 0.1, 0.2, 0.0, v3(0, 60, 179) * 0.3f / 255.0f, 10.);                    // water_bottom
    put(6, 1.0, 0.9, 0.1, 0.2, 0.5, v3(0, 43, 128) * 0.3f / 255.0f, 125.);                   // water_top
    put(7, 1.0, 0.9, 0.1, 0.0, 0.0, v3(136, 204, 0) * 0.3f / 255.0f, 10.);                   // island
    put(8, 1.0, 0.9, 0.1, 0.0, 0.0, v3(102, 51, 0) * 0.5f / 255.0f, 10.);                    // wood
    return 9;
}

// Model::Model, src/SceneObjects.cpp:211-255: "<nv> <nf>" then nv x "v x y z", nf x "f a b c"
// (1-based); coordinates are read as double, narrowed to float, then * scale + offset.
// Returns 0 and malloc'ed arrays (free with orx_free).
int orx_load_model(const char *path, float scale, const float *offset, float **verts, int *nv, int **idx, int *nt) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(RTX_E_INVALID, "orx_load_model: cannot open file");
    int vertNum = 0, faceNum = 0;
    if (fscanf(fp, "%d %d", &vertNum, &faceNum) != 2 || vertNum < 0 || faceNum < 0) {
        fclose(fp);
        return fail(RTX_E_INVALID, "orx_load_model: bad header");
    }
    float *v = (float *)malloc(sizeof(float) * 3 * (size_t)std::max(vertNum, 1));
    int *f = (int *)malloc(sizeof(int) * 3 * (size_t)std::max(faceNum, 1));
    for (int i = 0; i < vertNum; ++i) {
        char ch = 0;
        double x = 0, y = 0, z = 0;
        if (fscanf(fp, " %c %lf %lf %lf", &ch, &x, &y, &z) != 4) break;
        V3 p = v3((float)x, (float)y, (float)z) * scale + v3(offset[0], offset[1], offset[2]);
        v[3 * i] = p.x, v[3 * i + 1] = p.y, v[3 * i + 2] = p.z;
    }
    for (int i = 0; i < faceNum; ++i) {
        char ch = 0;
        int a = 0, b = 0, c = 0;
        if (fscanf(fp, " %c %d %d %d", &ch, &a, &b, &c) != 4) break;
        f[3 * i] = a - 1, f[3 * i + 1] = b - 1, f[3 * i + 2] = c - 1;
    }
    fclose(fp);
    *verts = v, *nv = vertNum, *idx = f, *nt = faceNum;
    return RTX_OK;
}
void orx_free(void *p) { free(p); }

// Camera of Scene::render, src/Scene.cpp:195-202, through glm::rotate
// (include/glm/ext/matrix_transform.inl:18-46) and mat3 * mat3 (type_mat3x3.inl).
void orx_default_camera(rtx_camera *cam) {
    const float fov = M_PI / 3.0f;
    float angleX = 15 * M_PI / float(180);
    float angleY = 15 * M_PI / float(180);
    auto rot = [](float angle, V3 v, float R[9]) {
        const float a = angle, c = cosf(a), s = sinf(a);
        V3 axis = normalize(v);
        V3 temp = axis * (1.0f - c);  // (T(1) - c) * axis
        float r00 = c + temp.x * axis.x, r01 = temp.x * axis.y + s * axis.z, r02 = temp.x * axis.z - s * axis.y;
        float r10 = temp.y * axis.x - s * axis.z, r11 = c + temp.y * axis.y, r12 = temp.y * axis.z + s * axis.x;
        float r20 = temp.z * axis.x + s * axis.y, r21 = temp.z * axis.y - s * axis.x, r22 = c + temp.z * axis.z;
        // Result[c] = m[0]*Rot[c][0] + m[1]*Rot[c][1] + m[2]*Rot[c][2] with m = identity; the upper
        // 3x3 is what mat3(mat4) keeps.
        const float rr[3][3] = {{r00, r01, r02}, {r10, r11, r12}, {r20, r21, r22}};
        for (int col = 0; col < 3; ++col)
            for (int row = 0; row < 3; ++row) {
                float m0 = row == 0 ? 1.0f : 0.0f, m1 = row == 1 ? 1.0f : 0.0f, m2 = row == 2 ? 1.0f : 0.0f;
                R[col * 3 + row] = m0 * rr[col][0] + m1 * rr[col][1] + m2 * rr[col][2];
            }
    };
    float RX[9], RY[9];
    rot(-angleX, v3(1, 0, 0), RX);
    rot(-angleY, v3(0, 1, 0), RY);
    // M = RY * RX, column c of the product: RY[0]*RX[c][0] + RY[1]*RX[c][1] + RY[2]*RX[c][2]
    for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row)
            cam->basis[col * 3 + row] =
                RY[0 * 3 + row] * RX[col * 3 + 0] + RY[1 * 3 + row] * RX[col * 3 + 1] + RY[2 * 3 + row] * RX[col * 3 + 2];
    cam->origin[0] = -7.5f, cam->origin[1] = 8.0f, cam->origin[2] = -5.0f;
    // tan(fov / 2.0f) at src/Scene.cpp:219 has a compile-time-constant argument; the reference's
    // compiler folds it to the correctly rounded value below (glibc's run-time tanf gives ...76).
    (void)fov;
    cam->tan_half_fov = 0x1.279a74p-1f;
    cam->add_origin = 1;
}

// The scene of Scene::load* (src/Scene.cpp:108-166) fed through the rtx_set_* calls above.
// mask bits: 1 spheres, 2 cubes, 4 loose triangles, 8 model, 16 islands.  Textures are set by
// the caller (water first if wanted, envmap always) before rtx_commit.
int orx_default_scene(rtx_ctx *c, unsigned mask, const char *bunny_obj_path) {
    float m9[81];
    orx_default_materials(m9);
    if (int e = rtx_set_materials(c, m9, 9)) return e;
    enum { GLASS = 1, MIRROR = 2, GOLD = 3, SILVER = 4, WATER_BOTTOM = 5, ISLAND = 7, WOOD = 8 };
    // Every category below is an std::unordered_set of heap pointers in the reference; with
    // libstdc++ and a handful of evenly spaced pointers such a set iterates in REVERSE insertion
    // order (each node opens a new bucket and is linked at the list head).  The order decides the
    // float summation order over the lights (src/Scene.cpp:64-78) and which object wins an exact
    // distance tie (strict `<` in the category loops), so arrays are listed back to front.
    const float lights[12] = {30, 20, 30, 1.7f, 35, 50, -25, 2.0f, -20, 20, 20, 1.5f};
    if (int e = rtx_set_lights(c, lights, 3)) return e;
    if (mask & 1) {
        const float cx[4] = {25.0f, -15.0f, -4, -6};
        const float cy[4] = {-4.0f + 4, -4.0f + 5, (float)-3.1, (float)-2.8};
        const float cz[4] = {-50.0f, -50.0f, -21, -22};
        const float r[4] = {4, 5, (float)0.7, 1};
        const int mat[4] = {GLASS, MIRROR, GLASS, SILVER};
        if (int e = rtx_set_spheres(c, cx, cy, cz, r, mat, 4)) return e;
    }
    if (mask & 2) {
        const float a[8][3] = {{-8, -4, -20}, {-7, -1.5f, -28}, {-7, 1.5f, -28}, {3, -4, -28}, {3, -4, -28},
                               {0, -4, -28}, {-3, -4, -28}, {-6, -4, -28}};
        const float b[8][3] = {{-2, -3.75f, -23}, {5, -0.5f, -29}, {5, 2.5f, -29}, {4, 3, -29}, {4, 3, -29},
                               {1, 3, -29}, {-2, 3, -29}, {-5, 3, -29}};
        const int mat[8] = {WOOD, WOOD, WOOD, WOOD, WOOD, WOOD, WOOD, WOOD};
        if (int e = rtx_set_boxes(c, &a[0][0], &b[0][0], mat, 8)) return e;
    }
    if (mask & 4) {
        const float t[4][9] = {{-8, -3.75f, -23, -8, -1, -23, -2, -3.75f, -23},
                               {-8, -3.75f, -23, -2, -3.75f, -23, -8, -1, -23},
                               {-8, -3.75f, -20, -8, -1, -23, -8, -3.75f, -23},
                               {-8, -3.75f, -20, -8, -3.75f, -23, -8, -1, -23}};
        const int mat[4] = {WOOD, WOOD, WOOD, WOOD};
        if (int e = rtx_set_triangles(c, &t[0][0], mat, 4)) return e;
    }
    if (mask & 8) {
        const float bias[3] = {4.0f, -7.0f + 1.4f, -23.0f};
        float *verts = nullptr;
        int *idx = nullptr;
        int nv = 0, nt = 0;
        if (int e = orx_load_model(bunny_obj_path, 45.0f, bias, &verts, &nv, &idx, &nt)) return e;
        int e = rtx_set_mesh(c, verts, nv, idx, nt, GOLD);
        free(verts), free(idx);
        if (e) return e;
    }
    if (mask & 16) {
        const float isl[16] = {15.0f, -100.0f, 15, 12, 25.0f, -50.0f, 10, 12, -15.0f, -50.0f, 12, 15, 0.0f, -25.0f, 12, 8};
        V3 lg = v3(136, 204, 0) * 0.3f / 255.0f, dg = v3(102, 153, 0) * 0.3f / 255.0f;
        const float l3[3] = {lg.x, lg.y, lg.z}, d3[3] = {dg.x, dg.y, dg.z};
        if (int e = rtx_set_islands(c, isl, 4, -4.0f, ISLAND, l3, d3)) return e;
    }
    return RTX_OK;
}

// bitmap_image::save_image, include/bitmap_image.hpp:435-483: 14-byte file header, 40-byte info
// header, rows bottom-up, each padded to 4 bytes; size_image masks the row size with 0xFFFC (:458).
int orx_save_bmp(const unsigned char *bgr, int W, int H, const char *path) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return fail(RTX_E_INVALID, "orx_save_bmp: cannot open file");
    auto w16 = [&](uint16_t v) { fwrite(&v, 2, 1, fp); };
    auto w32 = [&](uint32_t v) { fwrite(&v, 4, 1, fp); };
    uint32_t size_image = ((((uint32_t)W * 3) + 3) & 0x0000FFFC) * (uint32_t)H;
    w16(19778), w32(14 + 40 + size_image), w16(0), w16(0), w32(14 + 40);
    w32(40), w32((uint32_t)W), w32((uint32_t)H), w16(1), w16(24), w32(0), w32(size_image), w32(0), w32(0), w32(0), w32(0);
    unsigned padding = (4 - ((3 * (unsigned)W) % 4)) % 4;
    const char pad[4] = {0, 0, 0, 0};
    for (int i = 0; i < H; ++i) {
        fwrite(bgr + (size_t)W * 3 * (H - i - 1), 1, (size_t)W * 3, fp);
        fwrite(pad, 1, padding, fp);
    }
    fclose(fp);
    return RTX_OK;
}

}  // extern "C"
