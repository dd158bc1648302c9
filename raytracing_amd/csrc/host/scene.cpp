#include "scene.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <iostream>

#include "../../../include/rt_host.h"

namespace rt {

// ---- hard-coded scene, src/Scene.cpp:108-166 -------------------------------------------------
//
// Order matters.  The reference keeps every category in an std::unordered_set of heap pointers
// and the per-category loops accept only a strictly nearer hit, so among hits at exactly the same
// distance the object met first wins (for example a ray that starts inside both a fence post and a
// rail "hits" both at distance 0 and takes its normal from the first).  With libstdc++ a node that
// opens a new bucket is linked at the head of the element list, and the handful of equally spaced
// heap pointers of one category never share one of the set's 13 buckets, so each set iterates in
// REVERSE insertion order.  The arrays below are therefore filled back to front: element 0 is the
// object the reference inserted last.

Spheres Scene::loadSpheres() const {
    Spheres s;
    auto add = [&](float x, float y, float z, float r, const char *mat) {
        s.cx.push_back(x), s.cy.push_back(y), s.cz.push_back(z), s.r.push_back(r);
        s.mat.push_back(materials.index(mat));
    };
    add(25.0f, -4.0f + 4, -50.0f, 4, "glass");
    add(-15.0f, -4.0f + 5, -50.0f, 5, "mirror");
    add(-4, (float)-3.1, -21, (float)0.7, "glass");
    add(-6, (float)-2.8, -22, 1, "silver");
    return s;
}

Boxes Scene::loadCubes() const {
    Boxes b;
    const int wood = materials.index("wood");
    auto add = [&](float ax, float ay, float az, float bx, float by, float bz) {
        b.a.insert(b.a.end(), {ax, ay, az});
        b.b.insert(b.b.end(), {bx, by, bz});
        b.mat.push_back(wood);
    };
    // the slab under the wedge
    add(-8, -4, -20, -2, -3.75f, -23);
    // the fence: two rails, then five posts (one of them is listed twice in the reference)
    add(-7, -1.5f, -28, 5, -0.5f, -29);
    add(-7, 1.5f, -28, 5, 2.5f, -29);
    add(3, -4, -28, 4, 3, -29);
    add(3, -4, -28, 4, 3, -29);
    add(0, -4, -28, 1, 3, -29);
    add(-3, -4, -28, -2, 3, -29);
    add(-6, -4, -28, -5, 3, -29);
    return b;
}

Lights Scene::loadLights() const {
    // inserted as (-20,20,20;1.5), (35,50,-25;2.0), (30,20,30;1.7) (src/Scene.cpp:133-139) and
    // visited by calcLightning (src/Scene.cpp:64) in reverse, see above; the diffuse and specular
    // sums are float additions in visiting order
    Lights l;
    l.xyzi = {30, 20, 30, (float)1.7, 35, 50, -25, (float)2.0, -20, 20, 20, (float)1.5};
    return l;
}

Triangles Scene::loadTriangles() const {
    Triangles t;
    const int wood = materials.index("wood");
    // a wedge of two faces, each present with both windings (the test is single-sided)
    const float A[3] = {-8, -3.75f, -20}, B[3] = {-8, -3.75f, -23}, C[3] = {-8, -1, -23}, D[3] = {-2, -3.75f, -23};
    auto add = [&](const float *p, const float *q, const float *r) {
        t.v.insert(t.v.end(), p, p + 3);
        t.v.insert(t.v.end(), q, q + 3);
        t.v.insert(t.v.end(), r, r + 3);
        t.mat.push_back(wood);
    };
    add(B, C, D);
    add(B, D, C);
    add(A, C, B);
    add(A, B, C);
    return t;
}

Islands Scene::loadIslands() const {
    Islands is;
    is.cx_cz_a_b = {15.0f, -100.0f, 15, 12, 25.0f, -50.0f, 10, 12, -15.0f, -50.0f, 12, 15, 0.0f, -25.0f, 12, 8};
    is.plane_y = -4.0f;
    is.mat = materials.index("island");
    const float lg[3] = {136, 204, 0}, dg[3] = {102, 153, 0};
    for (int k = 0; k < 3; ++k) {
        is.light_green[k] = lg[k] * 0.3f / 255.0f;
        is.dark_green[k] = dg[k] * 0.3f / 255.0f;
    }
    return is;
}

bool Scene::loadModels(const std::string &bunny_path, Model &out) const {
    // bias = (4, -7 + 1.4, -23), scale 45, gold: src/Scene.cpp:153-154
    const float bias[3] = {4.0f, -7.0f + 1.4f, -23.0f};
    float *verts = nullptr;
    int *idx = nullptr;
    int nv = 0, nt = 0;
    out = Model();
    out.mat = materials.index("gold");
    if (rt_load_model(bunny_path.c_str(), 45.0f, bias, &verts, &nv, &idx, &nt) != 0) return false;
    out.verts.assign(verts, verts + 3 * (size_t)nv);
    out.idx.assign(idx, idx + 3 * (size_t)nt);
    rt_free(verts), rt_free(idx);
    return true;
}

void defaultCamera(rtx_camera *cam) {
    // rotateY * rotateX with both angles float(15 * pi / 180), built by glm::rotate
    // (src/Scene.cpp:197-201).  All inputs are compile-time constants in the reference, so the
    // matrix is a constant too; it is spelled out here in the exact bits its build produces rather
    // than recomputed through this build's cosf/sinf.  Column-major.
    static const float M[9] = {0x1.ee8dd4p-1f, 0x0p+0f,       0x1.0907dcp-2f,  //
                               0x1.126146p-4f, 0x1.ee8dd4p-1f, -0x1p-2f,       //
                               -0x1p-2f,       0x1.0907dcp-2f, 0x1.ddb3d6p-1f};
    for (int k = 0; k < 9; ++k) cam->basis[k] = M[k];
    cam->origin[0] = -7.5f, cam->origin[1] = 8.0f, cam->origin[2] = -5.0f;  // src/Scene.cpp:202
    // tan(fov / 2.0f) with fov = float(pi / 3), src/Scene.cpp:195,219, as the reference's compiler
    // folds it (correctly rounded); glibc's tanf returns the next float up for this argument.
    cam->tan_half_fov = 0x1.279a74p-1f;
    cam->add_origin = 1;  // src/Scene.cpp:220 adds the camera position to the direction
}

int Scene::upload(rtx_ctx *ctx, unsigned mask, const std::string &bunny_path, const unsigned char *water, int ww,
                  int wh, const unsigned char *sky, int sw, int sh) const {
    const std::vector<float> m9 = materials.flat();
    if (int e = rtx_set_materials(ctx, m9.data(), (int)materials.table().size())) return e;
    const Lights l = loadLights();
    if (int e = rtx_set_lights(ctx, l.xyzi.data(), (int)l.xyzi.size() / 4)) return e;
    if (mask & RT_MASK_SPHERES) {
        const Spheres s = loadSpheres();
        if (int e = rtx_set_spheres(ctx, s.cx.data(), s.cy.data(), s.cz.data(), s.r.data(), s.mat.data(), (int)s.mat.size())) return e;
    }
    if (mask & RT_MASK_CUBES) {
        const Boxes b = loadCubes();
        if (int e = rtx_set_boxes(ctx, b.a.data(), b.b.data(), b.mat.data(), (int)b.mat.size())) return e;
    }
    if (mask & RT_MASK_TRIANGLES) {
        const Triangles t = loadTriangles();
        if (int e = rtx_set_triangles(ctx, t.v.data(), t.mat.data(), (int)t.mat.size())) return e;
    }
    if (mask & RT_MASK_MODEL) {
        Model m;
        if (!loadModels(bunny_path, m)) std::cerr << "Failed to open '" << bunny_path << "' file." << std::endl;
        if (int e = rtx_set_mesh(ctx, m.verts.data(), (int)m.verts.size() / 3, m.idx.data(), (int)m.idx.size() / 3, m.mat)) return e;
    }
    if (mask & RT_MASK_ISLANDS) {
        const Islands is = loadIslands();
        if (int e = rtx_set_islands(ctx, is.cx_cz_a_b.data(), (int)is.cx_cz_a_b.size() / 4, is.plane_y, is.mat, is.light_green, is.dark_green)) return e;
    }
    // water plane y = -4.5, 10 units per texture tile, material water_bottom: src/Scene.cpp:38-46
    if (water)
        if (int e = rtx_set_water(ctx, water, ww, wh, -4.5f, 10.0f, materials.index("water_bottom"))) return e;
    // the sky lookup shifts rows by 100: src/SceneObjects.cpp:443
    return rtx_set_envmap(ctx, sky, sw, sh, 100);
}

int Scene::run(const Settings &settings) const {
    using clock = std::chrono::steady_clock;
    const auto t_start = clock::now();
    const bool stages = getenv("RT_TIMING") != nullptr;  // where the run's time goes, to stderr
    auto stage = [&](const char *what) {
        if (stages) std::cerr << "[rt] " << what << " done at " << std::chrono::duration<double>(clock::now() - t_start).count() << " s" << std::endl;
    };
    // (the reference's frame size is a compile-time constant; here it comes from the command line)
    if (settings.width <= 0 || settings.height <= 0 || (long long)settings.width * settings.height > (1ll << 30) || settings.depth < 0) {
        std::cout << "Error: bad frame size or depth (" << settings.width << " x " << settings.height << ", depth " << settings.depth << ")" << std::endl;
        return -1;
    }
    const std::string water_path = settings.resources + "/water.jpg";
    const std::string sky_path = settings.envmap.empty() ? settings.resources + "/map2.jpg" : settings.envmap;
    unsigned char *water = nullptr, *sky = nullptr;
    int ww = 0, wh = 0, wn = 0, sw = 0, sh = 0, sn = 0;
    if (rt_decode_jpeg(water_path.c_str(), &water, &ww, &wh, &wn) != 0 || wn != 3) {
        std::cout << "Error: can not load water texture: " << rt_host_last_error() << std::endl;
        rt_free(water);
        return -1;
    }
    std::cout << "[Current settings]" << std::endl;
    std::cout << "Output path: " << settings.out << std::endl;
    std::cout << "Scene number: " << settings.scene << std::endl;
    std::cout << "Threads: " << settings.threads << " (ignored: the pixel loop runs on " << settings.gpus << " GPU(s) from " << settings.device << ")" << std::endl;
    std::cout << "\nRendering image..." << std::endl;
    // The panorama (tens of megapixels) is decoded on a thread of its own while this one brings up the GPUs: the two
    // take about as long as each other, and the reference's run is dominated by exactly this loading (src/Scene.cpp:260).
    auto sky_decoded = std::async(std::launch::async, [&]() { return rt_decode_jpeg(sky_path.c_str(), &sky, &sw, &sh, &sn); });
    // one context per GPU, the same scene on each; one GPU is a group of one
    std::vector<int> ids;
    for (int k = 0; k < (settings.gpus > 1 ? settings.gpus : 1); ++k) ids.push_back(settings.device + k);
    rtx_group *group = nullptr;
    int rc = rtx_group_create(&group, ids.data(), (int)ids.size());
    stage("rtx_group_create");
    if (sky_decoded.get() != 0 || sn != 3) {
        // the reference prints this and then crashes in the first sky lookup (src/SceneObjects.cpp:424-427, 444)
        std::cout << "Error: can not load the environment map" << std::endl;
        rt_free(water), rt_free(sky);
        rtx_group_destroy(group);
        return -1;
    }
    for (int r = 0; rc == 0 && r < (int)ids.size(); ++r)
        rc = upload(rtx_group_ctx(group, r), RT_MASK_ALL, settings.resources + "/bunny.obj", water, ww, wh, sky, sw, sh);
    stage("panorama decoded, scene handed over");
    rt_free(water), rt_free(sky);
    if (rc == 0) rc = rtx_group_commit(group);
    stage("rtx_group_commit");
    // (page-locked: the frame leaves the GPU at the speed of the link)
    unsigned char *image = nullptr;
    if (rc == 0 && !(image = static_cast<unsigned char *>(rtx_host_alloc((size_t)settings.width * settings.height * 3)))) {
        std::cerr << "rtx error: no host memory for the frame" << std::endl;
        rtx_group_destroy(group);
        return -1;
    }
    if (rc == 0) {
        rtx_camera cam;
        defaultCamera(&cam);
        const auto t0 = clock::now();
        rc = rtx_group_render(group, &cam, settings.width, settings.height, settings.depth, image, nullptr);
        const auto t1 = clock::now();
        if (rc == 0) std::cout << "Done. Render time: " << std::chrono::duration<double>(t1 - t0).count() << std::endl;
    }
    if (rc != 0) std::cerr << "rtx error: " << (rtx_group_last_error()[0] ? rtx_group_last_error() : rtx_last_error()) << std::endl;
    stage("frame");
    if (rc == 0) rc = rt_save_bmp(image, settings.width, settings.height, settings.out.c_str());
    stage("file written");
    rtx_host_free(image);
    rtx_group_destroy(group);
    stage("rtx_group_destroy");
    std::cout << "Total time (including loading textures and models): "
              << std::chrono::duration<double>(clock::now() - t_start).count() << std::endl;
    return rc == 0 ? 0 : -1;
}

}  // namespace rt
