// Host-side scene description (counterpart of include/Scene.h + include/SceneObjects.h).
// The reference keeps heap objects in unordered sets and walks pointers per ray; here every
// category is a flat structure-of-arrays block that goes to the backend in one rtx_set_* call.
#pragma once
#include <string>
#include <vector>

#include "../../../include/rtx.h"
#include "material.hpp"

namespace rt {

struct Settings {  // include/Scene.h:8-15
    std::string out = "rt.bmp";
    unsigned scene = 1;
    unsigned threads = 8;
    // not in the reference (its values are compile-time constants there)
    int width = 800, height = 600, depth = 4;  // include/Scene.h:20-21, src/Scene.cpp:221
    std::string resources = "../resources";    // src/Scene.cpp:154,236
    std::string envmap;                        // default: <resources>/map2.jpg, include/SceneObjects.h:142
    int device = 0;
    int gpus = 1;  // rows of the frame interleaved over GPUs device, device + 1, ... (the reference: OpenMP threads)
};

struct Spheres {
    std::vector<float> cx, cy, cz, r;
    std::vector<int> mat;
};
struct Boxes {
    std::vector<float> a, b;  // opposite corners, xyz each
    std::vector<int> mat;
};
struct Triangles {
    std::vector<float> v;  // 9 floats per triangle
    std::vector<int> mat;
};
struct Lights {
    std::vector<float> xyzi;  // position + intensity, in the order calcLightning visits them
};
struct Islands {
    std::vector<float> cx_cz_a_b;
    float plane_y = -4.0f;  // hard-coded in Island::rayIntersect, src/SceneObjects.cpp:275
    int mat = 0;
    float light_green[3], dark_green[3];  // include/SceneObjects.h:105-106
};
struct Model {
    std::vector<float> verts;  // xyz, already scaled and offset
    std::vector<int> idx;      // zero-based
    int mat = 0;
};

class Scene {
public:
    const Materials materials;

    Spheres loadSpheres() const;      // src/Scene.cpp:108-115
    Boxes loadCubes() const;          // :117-131
    Lights loadLights() const;        // :133-139
    Triangles loadTriangles() const;  // :141-149
    Islands loadIslands() const;      // :159-166
    // :151-157; returns false (and leaves the model empty) when the file cannot be read, like the
    // reference, which only prints a message (src/SceneObjects.cpp:215-217)
    bool loadModels(const std::string &bunny_path, Model &out) const;

    // Hands the selected categories to the backend.  Textures: h x w x 3 bytes each.
    int upload(rtx_ctx *ctx, unsigned mask, const std::string &bunny_path, const unsigned char *water, int ww, int wh,
               const unsigned char *sky, int sw, int sh) const;

    // Scene::run + Scene::render (src/Scene.cpp:185-262) on the GPU backend.
    int run(const Settings &settings) const;
};

void defaultCamera(rtx_camera *cam);

}  // namespace rt
