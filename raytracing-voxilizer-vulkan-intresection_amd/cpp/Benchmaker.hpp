// Benchmaker.hpp -- the reference's N-run benchmark harness (src/hello_vulkan.h:172-241) for A/B runs against upstream:
// same class template, constructor and printed lines (typos included: upstream's output is the contract), whole
// milliseconds like upstream's duration_cast<milliseconds>.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <filesystem>
#include <numeric>
#include <vector>

#include "VoxelBuilder.hpp"
#include "octTree.hpp"

template <DerivedFromVoxelGrid T, bool UseOctree = false>
class Benchmaker final
{
    using Ms = std::chrono::milliseconds;
    using Clock = std::chrono::high_resolution_clock;

    std::vector<long long> m_buildMs, m_aabbMs;
    std::uint64_t m_bytes = 0;

    template <class Build>
    void timeOnce(Build&& build)
    {
        const auto t0 = Clock::now();
        auto structure = build();                 // VoxelBuilder<T>::buildVoxelGrid or Octree{...}
        const auto t1 = Clock::now();
        const auto boxes = structure.getAabbs();  // timed separately, as upstream does
        const auto t2 = Clock::now();
        (void)boxes;
        m_buildMs.push_back(std::chrono::duration_cast<Ms>(t1 - t0).count());
        m_aabbMs.push_back(std::chrono::duration_cast<Ms>(t2 - t1).count());
        m_bytes = structure.getMemoryUsageBytes();
    }

public:
    Benchmaker(const std::filesystem::path& path, float voxelSize, size_t runs)
    {
        for (size_t i = 0; i < runs; ++i) {
            if constexpr (UseOctree) {
                timeOnce([&] { return Octree{path, voxelSize}; });
            } else {
                VoxelBuilder<T> builder(path);  // the parse is outside the timed region (hello_vulkan.h:184)
                timeOnce([&] { return builder.buildVoxelGrid(voxelSize); });
            }
            std::printf("\n");
        }
        const double n = static_cast<double>(m_buildMs.size());
        const long long sb = std::accumulate(m_buildMs.begin(), m_buildMs.end(), 0ll);
        const long long sa = std::accumulate(m_aabbMs.begin(), m_aabbMs.end(), 0ll);
        std::printf("Voxel build took on avrage %gms\n", sb / n);
        std::printf("AABB build took on avrage %gms\n", sa / n);
        std::printf("Mem constium build took on avrage %llukb\n", static_cast<unsigned long long>(m_bytes));
        std::printf("Both together took an average build took on avrage %gms\n", (sa + sb) / n);
    }
};
