// facade_selftest.cpp -- exercises the C++ facade exactly as a reference caller would and writes the results to files the
// Python parity tests compare with the oracle:  facade_selftest <obj> <voxelsize> <outdir>
#include <cstdio>
#include <fstream>
#include <string>

#include "Benchmaker.hpp"
#include "VoxelBuilder.hpp"
#include "octTree.hpp"

static void dump(const std::string& file, const void* p, size_t bytes)
{
    std::ofstream f(file, std::ios::binary);
    f.write(reinterpret_cast<const char*>(p), (std::streamsize)bytes);
}

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const std::string path = argv[1], out = argv[3];
    const float vs = std::stof(argv[2]);
    int fails = 0;
    auto expect = [&](bool ok, const char* what) { if (!ok) { std::printf("FAIL: %s\n", what); ++fails; } };
    try {
        VoxelBuilder<VoxelGridBool> vb{std::filesystem::path(path)};
        VoxelGridBool g = vb.buildVoxelGrid(vs);
        const std::vector<Aabb> a = g.getAabbs();
        dump(out + "/bool.aabb", a.data(), a.size() * sizeof(Aabb));
        const auto w = g.words();
        dump(out + "/bool.words", w.data(), w.size() * 4);
        VoxelGridBool copy = g;  // grids are copy-constructible like the reference's
        expect(copy.getAabbs().size() == a.size(), "copy of a grid sees the same voxels");

        VoxelBuilder<VoxelGridBool, true> vbp{std::filesystem::path(path)};
        const std::vector<Aabb> ap = vbp.buildVoxelGrid(vs).getAabbs();
        dump(out + "/bool_parallel.aabb", ap.data(), ap.size() * sizeof(Aabb));

        VoxelBuilder<VoxelGridAABBstruct> vs2{std::filesystem::path(path)};
        VoxelGridAABBstruct ga = vs2.buildVoxelGrid(vs);
        const std::vector<Aabb> aa = ga.getAabbs();
        dump(out + "/aabbstruct.aabb", aa.data(), aa.size() * sizeof(Aabb));
        expect(ga.getMemoryUsageBytes() == 28 * ga.dimX() * ga.dimY() * ga.dimZ(), "AABBstruct memory = 28*N");

        VoxelBuilder<VoxelGridVec> vv{std::filesystem::path(path)};
        VoxelGridVec gv = vv.buildVoxelGrid(vs);
        const std::vector<Aabb> av = gv.getAabbs();
        dump(out + "/vec.aabb", av.data(), av.size() * sizeof(Aabb));
        expect(gv.getMemoryUsageBytes() == 24 * av.size(), "Vec memory = 24*hits");

        Octree tree{std::filesystem::path(path), vs};
        const std::vector<Aabb> ao = tree.getAabbs();
        dump(out + "/octree.aabb", ao.data(), ao.size() * sizeof(Aabb));
        const auto nd = tree.nodes();
        dump(out + "/octree.nodes", nd.data(), nd.size() * sizeof(Octree::Node));
        expect(tree.getMemoryUsageBytes() == 8 * tree.items().size() + 40 * nd.size(), "octree bytes = 8*items + 40*nodes");

        // VoxelGrid ctor + setVoxel + getCorrds + bounds errors
        VoxelGridBool h(5, 7, 3, 0.3f, vec3(0.5f, -1.25f, 3.0f));
        h.setVoxel(2, 3, 1);
        h.setVoxel(4, 6, 2, MaterialObj{});
        expect(h.getAabbs().size() == 2 && h.isOccupied(2, 3, 1) && !h.isOccupied(0, 0, 0), "setVoxel / isOccupied");
        const vec3 c = h.getCorrds(2, 3, 1);
        expect(c.x == 0.5f + (2.0f + 0.5f) * 0.3f && c.y == -1.25f + (3.0f + 0.5f) * 0.3f, "getCorrds formula");
        bool threw = false;
        try { h.setVoxel(5, 0, 0); } catch (const std::runtime_error& e) { threw = std::string(e.what()) == "Index out of bounds"; }
        expect(threw, "setVoxel out of bounds throws runtime_error(\"Index out of bounds\")");
        threw = false;
        try { (void)h.getCorrds(0, 7, 0); } catch (const std::runtime_error& e) { threw = std::string(e.what()) == "Index out of bounds"; }
        expect(threw, "getCorrds out of bounds throws");
        threw = false;
        try { VoxelBuilder<VoxelGridBool> bad{std::filesystem::path("/nonexistent/x.obj")}; } catch (const std::invalid_argument& e) { threw = std::string(e.what()) == "Path does not exist!"; }
        expect(threw, "missing path throws invalid_argument(\"Path does not exist!\")");
        // VoxelGrid::getVoxel (voxelgrid.hpp:66-72) returns m_voxel[map3dto1d(x,y,z)] -- per flavour:
        //  Bool: m_voxel is the WORD array, so a voxel index is used as a word index: (0,0,0) -> word 0, (3,0,0) -> word 3;
        //        5x7x3 = 105 voxels = 4 words, so any index >= 4 is past the array (the reference reads out of bounds there;
        //        here that is an exception, never a silent wrong value); coordinates outside the grid throw "Index out of bounds"
        {
            const auto hw = h.words();
            expect(hw.size() == 4, "5x7x3 grid has 4 words");
            expect(h.getVoxel(0, 0, 0) == hw[0] && h.getVoxel(1, 0, 0) == hw[1] && h.getVoxel(3, 0, 0) == hw[3], "Bool getVoxel = word[voxel index]");
            // (2,3,1) is voxel 2 + 5*(3 + 7*1) = 52 -> bit 20 of word 1; (4,6,2) is voxel 104 -> bit 8 of word 3
            expect(hw[1] == (1u << 20) && hw[3] == (1u << 8) && hw[0] == 0u && hw[2] == 0u, "setVoxel bit positions: word[idx/32] |= 1u << (idx%32)");
            bool t2 = false;
            try { (void)h.getVoxel(4, 0, 0); } catch (const std::out_of_range&) { t2 = true; }
            expect(t2, "Bool getVoxel past the word array throws instead of reading out of bounds");
            t2 = false;
            try { (void)h.getVoxel(5, 0, 0); } catch (const std::runtime_error& e) { t2 = std::string(e.what()) == "Index out of bounds"; }
            expect(t2, "getVoxel outside the grid throws runtime_error(\"Index out of bounds\")");
        }
        //  AABBstruct: m_voxel is the dense {min, max, isUsed} array: getVoxel is the voxel's own record
        {
            VoxelGridAABBstruct s3(5, 7, 3, 0.3f, vec3(0.5f, -1.25f, 3.0f));
            s3.setVoxel(2, 3, 1);
            const AabbInternal used = s3.getVoxel(2, 3, 1), unused = s3.getVoxel(0, 0, 0);
            const float cx = 0.5f + (2.0f + 0.5f) * 0.3f, hf = 0.5f * 0.3f;
            expect(used.isUsed && used.minimum.x == cx - hf && used.maximum.x == cx + hf, "AABBstruct getVoxel of a set voxel: c -/+ half, isUsed");
            expect(!unused.isUsed && unused.minimum.x == 0.f && unused.maximum.z == 0.f, "AABBstruct getVoxel of an unset voxel: zeros, !isUsed");
            const std::vector<Aabb> la = s3.getAabbs();
            expect(la.size() == 1 && la[0].minimum.x == used.minimum.x && la[0].maximum.y == used.maximum.y, "AABBstruct getAabbs = the used records");
        }
        //  Vec: m_voxel is the append-only list: getVoxel(x,y,z) is list element number map3dto1d(x,y,z)
        {
            VoxelGridVec v3(5, 7, 3, 0.3f, vec3(0.5f, -1.25f, 3.0f));
            v3.setVoxel(4, 6, 2);
            v3.setVoxel(2, 3, 1);
            v3.setVoxel(4, 6, 2);  // duplicates are kept
            const std::vector<Aabb> lv = v3.getAabbs();
            expect(lv.size() == 3, "Vec keeps one Aabb per setVoxel call");
            const Aabb e0 = v3.getVoxel(0, 0, 0), e1 = v3.getVoxel(1, 0, 0), e2 = v3.getVoxel(2, 0, 0);
            expect(e0.minimum.x == lv[0].minimum.x && e1.minimum.x == lv[1].minimum.x && e2.maximum.z == lv[2].maximum.z &&
                       e0.minimum.x == e2.minimum.x && e0.minimum.x != e1.minimum.x, "Vec getVoxel = list[voxel index], insertion order");
            bool t3 = false;
            try { (void)v3.getVoxel(3, 0, 0); } catch (const std::out_of_range&) { t3 = true; }
            expect(t3, "Vec getVoxel past the list throws");
        }
        h.addMatrialIfNeeded(3, MaterialObj{});
        expect(h.getMatrials().size() == 1 && h.getMatIdx().size() == 1 && h.getMatIdx()[0] == 0, "material helpers");
    } catch (const std::exception& e) {
        std::printf("FAIL: exception %s\n", e.what());
        return 1;
    }
    std::printf(fails ? "SELFTEST FAILED (%d)\n" : "SELFTEST OK\n", fails);
    return fails ? 1 : 0;
}
