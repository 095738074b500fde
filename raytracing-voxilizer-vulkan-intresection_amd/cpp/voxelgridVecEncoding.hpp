// voxelgridVecEncoding.hpp -- VoxelGridVec (reference: src/voxelgridVecEncoding.{hpp,cpp}): an append-only list of Aabb,
// one per setVoxel call, duplicates kept, in call order (triangle-major, then z, y, x when filled by VoxelBuilder).
#pragma once
#include "voxelgrid.hpp"

class VoxelGridVec final : public VoxelGrid<Aabb>
{
public:
    using VoxelType = Aabb;

    VoxelGridVec(size_t x, size_t y, size_t z, float voxelSize, vec3 org) : VoxelGrid(VX_GRID_VEC, x, y, z, voxelSize, org) {}
    VoxelGridVec(vxdetail::GridHandle h, const vx_grid_desc& d) : VoxelGrid(std::move(h), d) {}

    std::vector<Aabb> getAabbs() const noexcept override
    {
        try { return fetchAabbs(); } catch (...) { return {}; }
    }

    void setVoxel(size_t x, size_t y, size_t z, const MaterialObj& = MaterialObj{}) override { deviceSetVoxel(x, y, z); }

protected:
    // Reference quirk kept: m_voxel is the append-only list, so getVoxel(x,y,z) returns list element number
    // map3dto1d(x,y,z) (voxelgrid.hpp:66-72), not "the voxel at (x,y,z)".
    Aabb voxelAt(size_t i) const override
    {
        const std::vector<Aabb> a = fetchAabbs();
        if (i >= a.size()) throw std::out_of_range("VoxelGridVec::getVoxel: index past the appended list");
        return a[i];
    }
};
