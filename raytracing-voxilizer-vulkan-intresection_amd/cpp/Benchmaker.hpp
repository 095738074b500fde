// Benchmaker.hpp -- the reference's N-run benchmark harness (src/hello_vulkan.h:172-241), same printout, so the two builds
// can be A/B-compared by upstream users.  Times are whole milliseconds like upstream (duration_cast<milliseconds>).
#pragma once
#include <chrono>
#include <cstdio>
#include <filesystem>
#include <vector>

#include "VoxelBuilder.hpp"
#include "octTree.hpp"

template <DerivedFromVoxelGrid T, bool UseOctree = false>
class Benchmaker final
{
    const float m_voxelSize;
    const std::filesystem::path m_path;
    std::vector<std::chrono::milliseconds> m_VoxelBuildTime;
    std::vector<std::chrono::milliseconds> m_AABBBuildTime;
    uint64_t m_MemConsume = 0;

    void runBenachmark()
    {
        VoxelBuilder<T> voxelBuilder(m_path);
        const auto t0 = std::chrono::high_resolution_clock::now();
        T vox = voxelBuilder.buildVoxelGrid(m_voxelSize);
        const auto t1 = std::chrono::high_resolution_clock::now();
        const std::vector<Aabb> aabbs = vox.getAabbs();
        const auto t2 = std::chrono::high_resolution_clock::now();
        m_VoxelBuildTime.push_back(std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0));
        m_AABBBuildTime.push_back(std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1));
        m_MemConsume = vox.getMemoryUsageBytes();
    }

    void runBenachmarkOctree()
    {
        const auto t0 = std::chrono::high_resolution_clock::now();
        Octree tree{std::filesystem::path(m_path), m_voxelSize};
        const auto t1 = std::chrono::high_resolution_clock::now();
        const std::vector<Aabb> aabbs = tree.getAabbs();
        const auto t2 = std::chrono::high_resolution_clock::now();
        m_VoxelBuildTime.push_back(std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0));
        m_AABBBuildTime.push_back(std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1));
        m_MemConsume = tree.getMemoryUsageBytes();
    }

public:
    Benchmaker(const std::filesystem::path& path, float voxelSize, size_t runs) : m_voxelSize(voxelSize), m_path(path)
    {
        for (size_t i = 0; i < runs; i++) {
            if constexpr (UseOctree) runBenachmarkOctree();
            else runBenachmark();
        }
        long long sumVoxel = 0;
        for (const auto& d : m_VoxelBuildTime) { std::printf("\n"); sumVoxel += d.count(); }
        std::printf("Voxel build took on avrage %gms\n", static_cast<double>(sumVoxel) / m_VoxelBuildTime.size());
        long long sumAABB = 0;
        for (const auto& d : m_AABBBuildTime) sumAABB += d.count();
        std::printf("AABB build took on avrage %gms\n", static_cast<double>(sumAABB) / m_AABBBuildTime.size());
        std::printf("Mem constium build took on avrage %llukb\n", (unsigned long long)m_MemConsume);
        std::printf("Both together took an average build took on avrage %gms\n", static_cast<double>(sumAABB + sumVoxel) / m_AABBBuildTime.size());
    }
};
