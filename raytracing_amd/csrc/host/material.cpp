#include "material.hpp"

#include <iostream>

namespace rt {

namespace {
// colour given as 8-bit RGB times a float factor, divided by 255 -- the form
// `vec3(r, g, b) * k / 255.0f` of src/Material.cpp:13-16 (multiply first, then a true divide)
void scaled(float out[3], float r, float g, float b, float k) {
    out[0] = r * k / 255.0f, out[1] = g * k / 255.0f, out[2] = b * k / 255.0f;
}

Material make(double refr, double a0, double a1, double a2, double a3, double spec, const char *name) {
    Material m;
    // the reference's constructor takes `const float &` fed with double literals (include/Material.h:15)
    m.refractive = (float)refr;
    m.albedo[0] = (float)a0, m.albedo[1] = (float)a1, m.albedo[2] = (float)a2, m.albedo[3] = (float)a3;
    m.specular = (float)spec;
    m.name = name;
    return m;
}
}  // namespace

// src/Material.cpp:7-17, same order (primitives refer to materials by position)
Materials::Materials() {
    table_.push_back(Material());
    Material glass = make(1.5, 0.0, 0.5, 0.1, 0.8, 125., "glass");
    glass.diffuse[0] = (float)0.6, glass.diffuse[1] = (float)0.7, glass.diffuse[2] = (float)0.8;
    table_.push_back(glass);
    Material mirror = make(1.0, 0.0, 10.0, 0.8, 0.0, 1425., "mirror");
    mirror.diffuse[0] = mirror.diffuse[1] = mirror.diffuse[2] = 1.0f;
    table_.push_back(mirror);
    Material gold = make(1.0, 0.5, 3.0, 0.2, 0.0, 125, "gold");
    gold.diffuse[0] = 120 / 255.0f, gold.diffuse[1] = 85 / 255.0f, gold.diffuse[2] = 15 / 255.0f;
    table_.push_back(gold);
    Material silver = make(1.0, 0.5, 3.0, 0.2, 0.0, 125, "silver");
    silver.diffuse[0] = silver.diffuse[1] = silver.diffuse[2] = (float)0.3;
    table_.push_back(silver);
    Material water_bottom = make(1.0, 0.9, 0.1, 0.2, 0.0, 10., "water_bottom");
    scaled(water_bottom.diffuse, 0, 60, 179, 0.3f);
    table_.push_back(water_bottom);
    Material water_top = make(1.0, 0.9, 0.1, 0.2, 0.5, 125., "water_top");
    scaled(water_top.diffuse, 0, 43, 128, 0.3f);
    table_.push_back(water_top);
    Material island = make(1.0, 0.9, 0.1, 0.0, 0.0, 10., "island");
    scaled(island.diffuse, 136, 204, 0, 0.3f);
    table_.push_back(island);
    Material wood = make(1.0, 0.9, 0.1, 0.0, 0.0, 10., "wood");
    scaled(wood.diffuse, 102, 51, 0, 0.5f);
    table_.push_back(wood);
}

int Materials::index(const std::string &name) const {
    for (size_t i = 0; i < table_.size(); ++i)
        if (table_[i].name == name) return (int)i;
    std::cerr << "Failed to find material of type '" << name << "'. Load standard material." << std::endl;
    return 0;
}

std::vector<float> Materials::flat() const {
    std::vector<float> out;
    out.reserve(table_.size() * 9);
    for (const Material &m : table_) {
        out.push_back(m.refractive);
        for (float a : m.albedo) out.push_back(a);
        for (float d : m.diffuse) out.push_back(d);
        out.push_back(m.specular);
    }
    return out;
}

}  // namespace rt
