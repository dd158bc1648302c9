// Host-side material table (counterpart of include/Material.h and src/Material.cpp in the
// reference).  The device sees only the nine floats per material; names live on the host.
#pragma once
#include <string>
#include <vector>

namespace rt {

struct Material {
    float refractive = 1.0f;
    float albedo[4] = {1.0f, 0.0f, 0.0f, 0.0f};  // brightness, smoothness, reflectivity, refractivity
    float diffuse[3] = {0.0f, 0.0f, 0.0f};
    float specular = 0.0f;
    std::string name = "none";
};

class Materials {
public:
    Materials();
    // index of the material called `name`; unknown names report on stderr and give 0, the
    // default material (Materials::get, src/Material.cpp:19-27)
    int index(const std::string &name) const;
    const Material &operator[](const std::string &name) const { return table_[index(name)]; }
    const std::vector<Material> &table() const { return table_; }
    // n x {refractive, albedo[0..3], diffuse r,g,b, specular}
    std::vector<float> flat() const;

private:
    std::vector<Material> table_;
};

}  // namespace rt
