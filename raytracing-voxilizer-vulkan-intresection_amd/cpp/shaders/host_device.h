// host_device.h -- the two shared host/device definitions the voxel path uses (reference: src/shaders/host_device.h:117-124).
// The Vulkan descriptor plumbing of the reference header (bindings, push constants, ObjDesc ...) is out of scope.
#pragma once
#if __has_include(<glm/glm.hpp>)
#include <glm/glm.hpp>
using vec3 = glm::vec3;
#else
// glm is not part of this repository's toolchain: a minimal vec3 with the members the facade's signatures need.
namespace glm {
struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    constexpr vec3() = default;
    constexpr vec3(float s) : x(s), y(s), z(s) {}
    template <class A, class B, class C>
    constexpr vec3(A a, B b, C c) : x(static_cast<float>(a)), y(static_cast<float>(b)), z(static_cast<float>(c)) {}
    constexpr bool operator==(const vec3& o) const noexcept { return x == o.x && y == o.y && z == o.z; }
};
}  // namespace glm
using vec3 = glm::vec3;
#endif

struct Aabb  // host_device.h:117-121 -- 24 B, tightly packed; the element type of every getAabbs()
{
    vec3 minimum;
    vec3 maximum;
};
static_assert(sizeof(Aabb) == 24, "Aabb must be 6 packed floats");

#define KIND_SPHERE 0
#define KIND_CUBE 1  // hit kind reported by the intersection stage (raytrace.rint:66)
