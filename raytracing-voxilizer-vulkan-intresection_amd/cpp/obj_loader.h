// obj_loader.h -- MaterialObj, kept because setVoxel's signature takes one (reference: common/obj_loader.h:32-52).
// The reference's material plumbing on this path is commented out (VoxelBuilder.hpp:376-395, voxelgridBool.cpp:64):
// every voxel gets a default-constructed MaterialObj.  Field names, order and defaults are the reference's (callers
// aggregate-initialise and compare them); the hash only has to be consistent with operator==.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <initializer_list>

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

    // the reference's equality ignores ior and dissolve (obj_loader.h:47-51); kept so that de-duplication behaves the same
    bool operator==(const MaterialObj& o) const noexcept
    {
        const vec3* a[] = {&ambient, &diffuse, &specular, &transmittance, &emission};
        const vec3* b[] = {&o.ambient, &o.diffuse, &o.specular, &o.transmittance, &o.emission};
        for (int i = 0; i < 5; ++i)
            if (!(*a[i] == *b[i])) return false;
        return shininess == o.shininess && illum == o.illum && textureID == o.textureID;
    }
};

template <>
struct std::hash<MaterialObj> {
    std::size_t operator()(const MaterialObj& m) const noexcept
    {
        // FNV-1a over exactly the fields operator== looks at (with -0 folded onto +0 so equal materials hash equally)
        std::uint64_t h = 1469598103934665603ull;
        auto mix = [&h](float f) {
            if (f == 0.0f) f = 0.0f;
            std::uint32_t u;
            std::memcpy(&u, &f, 4);
            for (int k = 0; k < 4; ++k) { h ^= (u >> (8 * k)) & 0xFFu; h *= 1099511628211ull; }
        };
        for (const vec3* v : {&m.ambient, &m.diffuse, &m.specular, &m.transmittance, &m.emission}) { mix(v->x); mix(v->y); mix(v->z); }
        mix(m.shininess);
        mix(static_cast<float>(m.illum));
        mix(static_cast<float>(m.textureID));
        return static_cast<std::size_t>(h);
    }
};
