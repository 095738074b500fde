// voxelgridBool.hpp -- VoxelGridBool (reference: src/voxelgridBool.{hpp,cpp}): 1 bit per voxel in uint32 words, LSB first,
// voxel index x + X*(y + Y*z).  The words live in HBM (vx_grid of kind VX_GRID_BOOL).
#pragma once
#include "voxelgrid.hpp"

class VoxelGridBool final : public VoxelGrid<unsigned int>
{
public:
    using VoxelType = unsigned int;

    VoxelGridBool(size_t x, size_t y, size_t z, float voxelSize, vec3 org) : VoxelGrid(VX_GRID_BOOL, x, y, z, voxelSize, org) {}
    VoxelGridBool(vxdetail::GridHandle h, const vx_grid_desc& d) : VoxelGrid(std::move(h), d) {}

    // ascending word, ascending bit (voxelgridBool.cpp:18-52)
    std::vector<Aabb> getAabbs() const noexcept override
    {
        try { return fetchAabbs(); } catch (...) { return {}; }
    }

    // word[idx/32] |= 1u << (idx%32) (voxelgridBool.cpp:54-68)
    void setVoxel(size_t x, size_t y, size_t z, const MaterialObj& = MaterialObj{}) override { deviceSetVoxel(x, y, z); }

    // host copy of the bitmask words
    std::vector<unsigned int> words() const
    {
        vx_grid_desc d;
        vxdetail::check(vx_grid_describe(m_grid.get(), &d));
        std::vector<unsigned int> w(d.num_words);
        vxdetail::check(vx_grid_bitmask(m_grid.get(), w.data(), w.size()));
        return w;
    }

protected:
    // Reference quirk kept (voxelgrid.hpp:66-72 + SURVEY a15): getVoxel indexes the WORD array with a VOXEL index, i.e. it
    // returns word number map3dto1d(x,y,z) -- out of range for most voxels.  Here: that word if it exists, else an
    // exception instead of the reference's out-of-bounds read.  Use isOccupied() for the bit.
    unsigned int voxelAt(size_t i) const override
    {
        const std::vector<unsigned int> w = words();
        if (i >= w.size()) throw std::out_of_range("VoxelGridBool::getVoxel: voxel index used as word index is past the word array");
        return w[i];
    }
};
