// octTree.hpp -- Octree with the reference's interface (src/octTree.hpp:487-528), built on an MI355X through libvoxhip.so:
// SAT voxelization -> one Morton code per hit (duplicates kept) -> device sort -> flat pre-order node array.
#pragma once
#include <voxhip.h>

#include <cstdio>
#include <filesystem>
#include <memory>
#include <vector>

#include "VoxelBuilder.hpp"

class Octree final
{
public:
    using MortonCode = std::uint64_t;
    struct Item { MortonCode morton; };                      // octTree.hpp:37-41
    using Node = vx_octree_node;                             // {children[8], start, count}, 40 B (octTree.hpp:251-277)
    static constexpr std::uint32_t INVALID_INDEX = 0xFFFFFFFFu;

    explicit Octree(const std::filesystem::path& path, float voxSize, size_t maxItemsPerLeaf = 16)
    {
        if (!std::filesystem::exists(path)) { throw std::invalid_argument("Path does not exist!"); }  // octTree.hpp:300-302
        vx_mesh* m = nullptr;
        vxdetail::check(vx_mesh_load_obj(path.string().c_str(), &m));
        std::unique_ptr<vx_mesh, vxdetail::MeshDeleter> mesh(m);
        build(mesh.get(), voxSize, maxItemsPerLeaf);
    }
    Octree(const float* xyz, size_t numVertices, const int* triIndices, size_t numTriangles, float voxSize, size_t maxItemsPerLeaf = 16)
    {
        vx_mesh* m = nullptr;
        vxdetail::check(vx_mesh_from_arrays(xyz, numVertices, triIndices, numTriangles, &m));
        std::unique_ptr<vx_mesh, vxdetail::MeshDeleter> mesh(m);
        build(mesh.get(), voxSize, maxItemsPerLeaf);
    }

    std::vector<Aabb> getAabbs() const noexcept  // octTree.hpp:502-510: items in sorted order, duplicates included
    {
        try {
            uint64_t n = 0;
            vxdetail::check(vx_octree_aabbs(m_o.get(), nullptr, 0, &n));
            std::vector<Aabb> ret(n);
            if (n) vxdetail::check(vx_octree_aabbs(m_o.get(), reinterpret_cast<vx_aabb*>(ret.data()), n, &n));
            return ret;
        } catch (...) { return {}; }
    }

    size_t getMemoryUsageBytes() const noexcept { return static_cast<size_t>(vx_octree_bytes(m_o.get())); }  // 8*items + 40*nodes

    std::vector<Item> items() const
    {
        std::vector<Item> it(vx_octree_num_items(m_o.get()));
        vxdetail::check(vx_octree_items(m_o.get(), reinterpret_cast<uint64_t*>(it.data()), it.size()));
        return it;
    }
    std::vector<Node> nodes() const
    {
        std::vector<Node> nd(vx_octree_num_nodes(m_o.get()));
        vxdetail::check(vx_octree_nodes(m_o.get(), nd.data(), nd.size()));
        return nd;
    }
    vx_octree* handle() const noexcept { return m_o.get(); }

    Octree(const Octree&) = delete;
    Octree& operator=(const Octree&) = delete;
    Octree(Octree&&) noexcept = default;
    Octree& operator=(Octree&&) noexcept = default;

private:
    struct Del { void operator()(vx_octree* o) const noexcept { vx_octree_free(o); } };
    std::unique_ptr<vx_octree, Del> m_o;

    void build(vx_mesh* mesh, float voxSize, size_t maxItems)
    {
        vx_octree* o = nullptr;
        vxdetail::check(vx_octree_build(mesh, voxSize, maxItems, nullptr, &o));  // VX_ERR_MORTON_BITS -> runtime_error, octTree.hpp:583-585
        m_o.reset(o);
        if (!vxdetail::quiet()) {
            // octTree.hpp:568-569, :798-808
            std::printf("Voxel size: %s\n", vxdetail::fmt(voxSize).c_str());
            std::printf("Total triangles processed: %zu\n", vx_mesh_num_triangles(mesh));
            std::printf("Total voxels inserted (before tree build): %zu\n", (size_t)vx_octree_num_items(o));
            std::printf("Total octree nodes: %zu\n", (size_t)vx_octree_num_nodes(o));
        }
    }
};
