// voxilizer.cpp -- the reference's command line for this path:  voxilizer <Path to obj file> <Voxlesize>
// (README.md:57, main.cpp:163 -> HelloVulkan::createAABB, hello_vulkan.cpp:669-699), without the Vulkan window.
// It performs exactly createAABB's call sequence -- VoxelBuilder<VoxelGridBool>{path}, buildVoxelGrid(vs) [timed],
// getAabbs() [timed] -- and prints the reference's three result lines, then throughput figures.
// Optional flags after the two positionals:
//   --grid bool|aabbstruct|vec|octree   grid flavour (default bool, the app's default: hello_vulkan.cpp:677)
//   --parallel                          reproduce the threaded driver's SAT (VoxelBuilder<T,true>)
//   --rays WxH                          also trace WxH primary rays from the reference camera (main.cpp:92, rgen:41-51)
//   --dump FILE                         write the AABB list as raw 24-byte records
//   --bench RUNS                        Benchmaker<T>{path, vs, RUNS} printout (hello_vulkan.h:172-241)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

#include "Benchmaker.hpp"
#include "VoxelBuilder.hpp"
#include "octTree.hpp"

namespace {
using Clock = std::chrono::high_resolution_clock;

void dump(const std::string& file, const std::vector<Aabb>& a)
{
    if (file.empty()) return;
    std::ofstream f(file, std::ios::binary);
    f.write(reinterpret_cast<const char*>(a.data()), (std::streamsize)(a.size() * sizeof(Aabb)));
}

template <class T, bool P>
int run_grid(const std::string& path, float vs, const std::string& dumpFile, const char* label)
{
    VoxelBuilder<T, P> voxelBuilder{std::filesystem::path(path)};
    const auto t0 = Clock::now();
    T vox = voxelBuilder.buildVoxelGrid(vs);
    const auto t1 = Clock::now();
    const std::vector<Aabb> aabbs = vox.getAabbs();
    const auto t2 = Clock::now();
    const auto msBuild = std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count();
    const auto msAabb = std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count();
    std::printf("Voxel build took %lldms\n", (long long)msBuild);                                   // hello_vulkan.cpp:686
    std::printf("Aabb build took %lldms\n", (long long)msAabb);                                     // :687
    std::printf("Total usage of the VoxelGridAABBstruct is %zu\n", vox.getMemoryUsageBytes());      // :688 (label as upstream)
    const double sb = std::chrono::duration<double>(t1 - t0).count(), sa = std::chrono::duration<double>(t2 - t1).count();
    const double cells = (double)vox.dimX() * vox.dimY() * vox.dimZ();
    std::printf("[voxhip] %s: %zu AABBs, %.1f Mvoxels/s build (host wall, incl. launch+sync), %.1f M AABBs/s getAabbs (incl. D2H copy)\n", label,
                aabbs.size(), cells / sb / 1e6, aabbs.size() / (sa > 0 ? sa : 1e-9) / 1e6);
    dump(dumpFile, aabbs);
    return 0;
}
}  // namespace

int main(int argc, char** argv)
{
    if (argc < 3) {  // the reference reads argv[1], argv[2] unchecked (main.cpp:80,163)
        std::fprintf(stderr, "usage: %s <Path to obj file> <Voxlesize> [--grid bool|aabbstruct|vec|octree] [--parallel] [--dump FILE] [--bench RUNS]\n",
                     argv[0]);
        return 2;
    }
    const std::string path = argv[1];
    float vs = 0.f;
    try { vs = std::stof(argv[2]); } catch (const std::exception&) { std::fprintf(stderr, "invalid voxel size '%s'\n", argv[2]); return 2; }
    std::string grid = "bool", dumpFile;
    bool parallel = false;
    long benchRuns = 0;
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--grid") && i + 1 < argc) grid = argv[++i];
        else if (!std::strcmp(argv[i], "--parallel")) parallel = true;
        else if (!std::strcmp(argv[i], "--dump") && i + 1 < argc) dumpFile = argv[++i];
        else if (!std::strcmp(argv[i], "--bench") && i + 1 < argc) benchRuns = std::atol(argv[++i]);
        else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    try {
        if (benchRuns > 0) {
            if (grid == "octree") Benchmaker<VoxelGridBool, true>{std::filesystem::path(path), vs, (size_t)benchRuns};
            else if (grid == "vec") Benchmaker<VoxelGridVec>{std::filesystem::path(path), vs, (size_t)benchRuns};
            else if (grid == "aabbstruct") Benchmaker<VoxelGridAABBstruct>{std::filesystem::path(path), vs, (size_t)benchRuns};
            else Benchmaker<VoxelGridBool>{std::filesystem::path(path), vs, (size_t)benchRuns};
            return 0;
        }
        if (grid == "octree") {
            // the commented-out alternative in createAABB (hello_vulkan.cpp:690-697)
            const auto t0 = Clock::now();
            Octree tree{std::filesystem::path(path), vs};
            const auto t1 = Clock::now();
            std::printf("Total usage of the Octree is %zu\n", tree.getMemoryUsageBytes());
            std::printf("Voxel build took %lldms\n", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count());
            const std::vector<Aabb> aabbs = tree.getAabbs();
            std::printf("[voxhip] octree: %zu AABBs\n", aabbs.size());
            dump(dumpFile, aabbs);
            return 0;
        }
        if (grid == "bool") return parallel ? run_grid<VoxelGridBool, true>(path, vs, dumpFile, "VoxelGridBool") : run_grid<VoxelGridBool, false>(path, vs, dumpFile, "VoxelGridBool");
        if (grid == "aabbstruct") return parallel ? run_grid<VoxelGridAABBstruct, true>(path, vs, dumpFile, "VoxelGridAABBstruct") : run_grid<VoxelGridAABBstruct, false>(path, vs, dumpFile, "VoxelGridAABBstruct");
        if (grid == "vec") return parallel ? run_grid<VoxelGridVec, true>(path, vs, dumpFile, "VoxelGridVec") : run_grid<VoxelGridVec, false>(path, vs, dumpFile, "VoxelGridVec");
        std::fprintf(stderr, "unknown grid flavour %s\n", grid.c_str());
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
