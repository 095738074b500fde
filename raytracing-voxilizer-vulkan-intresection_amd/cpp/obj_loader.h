// obj_loader.h -- MaterialObj, kept because setVoxel's signature takes one (reference: common/obj_loader.h:32-52,87-115).
// The reference's material plumbing on this path is commented out (VoxelBuilder.hpp:376-395, voxelgridBool.cpp:64):
// every voxel gets a default-constructed MaterialObj.
#pragma once
#include <cstddef>
#include <functional>
#include "shaders/host_device.h"

struct MaterialObj {
    vec3 ambient = vec3(0.1f, 0.1f, 0.1f);
    vec3 diffuse = vec3(1, 1, 0);
    vec3 specular = vec3(1.0f, 1.0f, 1.0f);
    vec3 transmittance = vec3(0.0f, 0.0f, 0.0f);
    vec3 emission = vec3(0.0f, 0.0f, 0.10f);
    float shininess = 0.f;
    float ior = 1.0f;
    float dissolve = 1.f;
    int illum = 0;
    int textureID = -1;

    bool operator==(const MaterialObj& o) const noexcept
    {
        return ambient == o.ambient && diffuse == o.diffuse && specular == o.specular && transmittance == o.transmittance &&
               emission == o.emission && shininess == o.shininess && illum == o.illum && textureID == o.textureID;
    }
};

namespace vxdetail {
inline void hash_combine(std::size_t& seed, std::size_t h) { seed ^= h + 0x9e3779b9 + (seed << 6) + (seed >> 2); }
inline std::size_t hash_vec3(const vec3& v)
{
    std::size_t h = 0;
    hash_combine(h, std::hash<float>{}(v.x));
    hash_combine(h, std::hash<float>{}(v.y));
    hash_combine(h, std::hash<float>{}(v.z));
    return h;
}
}  // namespace vxdetail

namespace std {
template <>
struct hash<MaterialObj> {
    std::size_t operator()(const MaterialObj& m) const noexcept
    {
        std::size_t h = 0;
        vxdetail::hash_combine(h, vxdetail::hash_vec3(m.ambient));
        vxdetail::hash_combine(h, vxdetail::hash_vec3(m.diffuse));
        vxdetail::hash_combine(h, vxdetail::hash_vec3(m.specular));
        vxdetail::hash_combine(h, vxdetail::hash_vec3(m.transmittance));
        vxdetail::hash_combine(h, vxdetail::hash_vec3(m.emission));
        vxdetail::hash_combine(h, std::hash<float>{}(m.shininess));
        vxdetail::hash_combine(h, std::hash<float>{}(m.ior));
        vxdetail::hash_combine(h, std::hash<float>{}(m.dissolve));
        vxdetail::hash_combine(h, std::hash<int>{}(m.illum));
        vxdetail::hash_combine(h, std::hash<int>{}(m.textureID));
        return h;
    }
};
}  // namespace std
