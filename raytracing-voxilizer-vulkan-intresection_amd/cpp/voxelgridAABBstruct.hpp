// voxelgridAABBstruct.hpp -- VoxelGridAABBstruct (reference: src/voxelgridAABBstruct.{hpp,cpp}).
// The reference keeps a dense array of {min, max, isUsed} (28 B per voxel) and getAabbs() filters isUsed in index order,
// which yields byte-for-byte the VoxelGridBool list.  Here the dense array is never materialised:
// occupancy is the same 1-bit mask in HBM and the {min,max} of a voxel is computed on demand with the reference's float
// recipe.  getMemoryUsageBytes() still reports the reference's figure, 28 * X*Y*Z.
#pragma once
#include "voxelgrid.hpp"

namespace {
struct AabbInternal
{
    vec3 minimum = {0.f, 0.f, 0.f};
    vec3 maximum = {0.f, 0.f, 0.f};
    bool isUsed = false;
};
}  // namespace

class VoxelGridAABBstruct final : public VoxelGrid<AabbInternal>
{
public:
    using VoxelType = AabbInternal;

    VoxelGridAABBstruct(size_t x, size_t y, size_t z, float voxelSize, vec3 org = {0.f, 0.f, 0.f})
        : VoxelGrid(VX_GRID_AABBSTRUCT, x, y, z, voxelSize, org) {}
    VoxelGridAABBstruct(vxdetail::GridHandle h, const vx_grid_desc& d) : VoxelGrid(std::move(h), d) {}

    std::vector<Aabb> getAabbs() const noexcept override
    {
        try { return fetchAabbs(); } catch (...) { return {}; }
    }

    void setVoxel(size_t x, size_t y, size_t z, const MaterialObj& = MaterialObj{}) override { deviceSetVoxel(x, y, z); }

protected:
    AabbInternal voxelAt(size_t i) const override
    {
        const size_t x = i % m_x, y = (i / m_x) % m_y, z = i / (m_x * m_y);
        AabbInternal r;
        if (isOccupied(x, y, z)) {
            const vec3 c = getCorrds(x, y, z);
            const float half = 0.5f * m_voxelSize;  // voxelgridAABBstruct.cpp:36-42
            r.minimum = vec3(c.x - half, c.y - half, c.z - half);
            r.maximum = vec3(c.x + half, c.y + half, c.z + half);
            r.isUsed = true;
        }
        return r;
    }
};
