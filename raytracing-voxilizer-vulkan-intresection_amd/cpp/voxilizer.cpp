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
//   --render FILE.ppm [--size WxH]      the reference's picture of the voxels without Vulkan: primary rays from the reference
//                                       camera (main.cpp:92, raytrace.rgen:41-51), cube normals + Lambert + shadow ray as in
//                                       raytrace2.rchit:53-137 with the default material and light (hello_vulkan.h:84-90), miss
//                                       colour raytrace.rmiss:37, gamma post.frag:36.  Rays run on the GPU, the per-pixel shading
//                                       arithmetic (display, not the hot path) on the host.  --camera-dump FILE writes the two 4x4
//                                       matrices used (column-major float32), for comparisons.
//   --gpus N [--logical]                spread the build over N GPUs of this node (word shards + peer copies, vx_voxelize_multi);
//                                       --logical maps all N ranks onto device 0 (rehearsal on a box with fewer GPUs)
//   --materials                         switch on the reference's commented-out material plumbing (usemtl / mtllib -> per-voxel
//                                       material ids; VoxelBuilder.hpp:375-395): --render shades with them, --dump-materials FILE writes
//                                       getMatIdx() as int16
#include <chrono>
#include <cmath>
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

struct V3 { float x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 norm(V3 a) { const float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }

// viewInverse / projInverse as the reference uploads them (hello_vulkan.cpp:69-77): lookAt RH, perspectiveRH_ZO, [1][1] *= -1
void camera(float vi[16], float pi[16], float aspect)
{
    const V3 eye{6.16636f, 2.42256f, -3.15471f}, ctr{0.f, 1.f, 0.f}, up{0.f, 1.f, 0.f};  // main.cpp:92
    const V3 f = norm(ctr - eye), s = norm(cross(f, up)), u = cross(s, f);
    const float m[16] = {s.x, s.y, s.z, 0, u.x, u.y, u.z, 0, -f.x, -f.y, -f.z, 0, eye.x, eye.y, eye.z, 1};  // columns: s, u, -f, eye
    for (int i = 0; i < 16; ++i) vi[i] = m[i];
    const float fov = 60.0f * 3.14159265358979f / 180.0f;  // nvpro CameraManip default (third-party; unpinned in the tree)
    const float t = std::tan(fov * 0.5f), zn = 0.1f, zf = 1000.0f;
    const float a = 1.0f / (aspect * t), b = -1.0f / t, c = zf / (zn - zf), d = -(zf * zn) / (zf - zn);
    for (int i = 0; i < 16; ++i) pi[i] = 0.f;
    pi[0] = 1.0f / a; pi[5] = 1.0f / b; pi[11] = 1.0f / d; pi[14] = -1.0f; pi[15] = c / d;  // column-major inverse of the projection
}

struct RenderOpts { std::vector<MaterialObj> materials; std::vector<int16_t> matIdx; std::string cameraDump; };

int render(vx_grid* grid, const std::string& file, uint32_t W, uint32_t H, const RenderOpts& ro)
{
    float vi[16], pi[16];
    camera(vi, pi, (float)W / (float)H);
    if (!ro.cameraDump.empty()) {
        std::ofstream cf(ro.cameraDump, std::ios::binary);
        cf.write(reinterpret_cast<const char*>(vi), 64);
        cf.write(reinterpret_cast<const char*>(pi), 64);
    }
    const size_t n = (size_t)W * H;
    std::vector<float> t(n), nrm(3 * n), rays(6 * n), tmaxs(n);
    std::vector<uint32_t> prim(n);
    std::vector<uint8_t> shadowed(n);
    vx_trace_args a{};
    a.view_inverse = vi; a.proj_inverse = pi; a.width = W; a.height = H; a.tmin = 0.001f; a.tmax = 10000.0f;  // rgen:50-51
    a.t = t.data(); a.prim = prim.data(); a.normal = nrm.data();
    vxdetail::check(vx_trace_ex(grid, &a));
    // shadow rays from the hit points toward the point light (rchit:76-122)
    const V3 light{10.f, 55.f, 8.f};  // hello_vulkan.h:86
    const float intensity = 1000.f;   // :88
    const V3 org{vi[12], vi[13], vi[14]};
    std::vector<V3> dirs(n);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t px = (uint32_t)(i % W), py = (uint32_t)(i / W);
        const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H, dx = u * 2.f - 1.f, dy = v * 2.f - 1.f;
        const V3 tg = norm(V3{pi[0] * dx + pi[4] * dy + pi[8] + pi[12], pi[1] * dx + pi[5] * dy + pi[9] + pi[13], pi[2] * dx + pi[6] * dy + pi[10] + pi[14]});
        dirs[i] = V3{vi[0] * tg.x + vi[4] * tg.y + vi[8] * tg.z, vi[1] * tg.x + vi[5] * tg.y + vi[9] * tg.z, vi[2] * tg.x + vi[6] * tg.y + vi[10] * tg.z};
        const V3 wp = org + dirs[i] * (t[i] > 0 ? t[i] : 0.f);
        const V3 l = light - wp;
        const float dist = std::sqrt(dot(l, l));
        const V3 L = l * (1.0f / dist);
        rays[6 * i + 0] = wp.x; rays[6 * i + 1] = wp.y; rays[6 * i + 2] = wp.z;
        rays[6 * i + 3] = L.x; rays[6 * i + 4] = L.y; rays[6 * i + 5] = L.z;
        tmaxs[i] = dist;
    }
    vx_trace_args sa{};
    sa.rays = rays.data(); sa.num_rays = n; sa.tmin = 0.001f; sa.tmax = 10000.0f; sa.tmax_per_ray = tmaxs.data(); sa.any_hit = 1;
    sa.shadowed = shadowed.data();
    vxdetail::check(vx_trace_ex(grid, &sa));
    const MaterialObj defmat{};  // the single default material createAABB uploads (hello_vulkan.cpp:701-702)
    std::vector<unsigned char> img(3 * n);
    size_t hits = 0;
    for (size_t i = 0; i < n; ++i) {
        float c[3] = {0.8f, 0.8f, 0.8f};  // rmiss:37 with the white clear colour of main.cpp:184
        if (t[i] > 0) {
            ++hits;
            const V3 N{nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]}, L{rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]};
            // matIndices.i[gl_PrimitiveID] -> materials.m[matIdx] (rchit:92-94); without --materials: index 0 of the one default material
            const MaterialObj& mat = (!ro.matIdx.empty() && prim[i] < ro.matIdx.size() && (size_t)ro.matIdx[prim[i]] < ro.materials.size())
                                         ? ro.materials[(size_t)ro.matIdx[prim[i]]] : defmat;
            const float li = intensity / (tmaxs[i] * tmaxs[i]);                 // rchit:85
            const float dnl = std::fmax(dot(N, L), 0.0f);                       // wavefront.glsl:25
            float diff[3] = {mat.diffuse.x * dnl, mat.diffuse.y * dnl, mat.diffuse.z * dnl};
            if (mat.illum >= 1) { diff[0] += mat.ambient.x; diff[1] += mat.ambient.y; diff[2] += mat.ambient.z; }
            float att = 0.3f, spec[3] = {0.f, 0.f, 0.f};                        // rchit:99-133
            if (dot(N, L) > 0 && !shadowed[i]) {
                att = 1.0f;
                if (mat.illum >= 2) {                                           // computeSpecular, wavefront.glsl:32-48
                    const float kPi = 3.14159265f, kSh = std::fmax(mat.shininess, 4.0f);
                    const float kE = (2.0f + kSh) / (2.0f * kPi);
                    const V3 V = norm(dirs[i] * -1.0f);
                    const V3 I = L * -1.0f;                                     // reflect(-L, N) = I - 2 dot(N, I) N
                    const V3 Rr = I - N * (2.0f * dot(N, I));
                    const float sp = kE * std::pow(std::fmax(dot(V, Rr), 0.0f), kSh);
                    spec[0] = mat.specular.x * sp; spec[1] = mat.specular.y * sp; spec[2] = mat.specular.z * sp;
                }
            }
            for (int k = 0; k < 3; ++k) c[k] = li * att * (diff[k] + spec[k]);
        }
        for (int k = 0; k < 3; ++k) {
            const float g = std::pow(std::fmin(std::fmax(c[k], 0.f), 1.f), 1.0f / 2.2f);  // post.frag:36
            img[3 * i + k] = (unsigned char)std::lround(g * 255.0f);
        }
    }
    std::ofstream f(file, std::ios::binary);
    f << "P6\n" << W << " " << H << "\n255\n";
    f.write(reinterpret_cast<const char*>(img.data()), (std::streamsize)img.size());
    std::printf("[voxhip] rendered %ux%u to %s: %zu of %zu primary rays hit a voxel\n", W, H, file.c_str(), hits, n);
    return 0;
}

template <class T, bool P>
int run_grid(const std::string& path, float vs, const std::string& dumpFile, const char* label, const std::string& renderFile = "",
             uint32_t rw = 1280, uint32_t rh = 720, bool materials = false, const std::string& matDump = "", const std::string& cameraDump = "",
             const std::vector<int>& devices = {})
{
    VoxelBuilder<T, P> voxelBuilder{std::filesystem::path(path)};
    voxelBuilder.withMaterials(materials);
    voxelBuilder.withDevices(devices);
    if (devices.size() > 1) std::printf("[voxhip] build sharded over %zu ranks (first device %d)\n", devices.size(), devices[0]);
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
    RenderOpts ro;
    ro.cameraDump = cameraDump;
    if (materials) {
        ro.materials = vox.getMatrials();
        ro.matIdx = vox.getMatIdx();
        std::printf("[voxhip] materials: %zu distinct, %zu per-voxel ids\n", ro.materials.size(), ro.matIdx.size());
        if (!matDump.empty()) {
            std::ofstream f(matDump, std::ios::binary);
            f.write(reinterpret_cast<const char*>(ro.matIdx.data()), (std::streamsize)(ro.matIdx.size() * sizeof(int16_t)));
        }
    }
    if (!renderFile.empty()) return render(vox.handle(), renderFile, rw, rh, ro);
    return 0;
}
}  // namespace

int main(int argc, char** argv)
{
    if (argc < 3) {  // the reference reads argv[1], argv[2] unchecked (main.cpp:80,163)
        std::fprintf(stderr, "usage: %s <Path to obj file> <Voxlesize> [--grid bool|aabbstruct|vec|octree] [--parallel] [--dump FILE] [--bench RUNS] [--render FILE.ppm [--size WxH] [--camera-dump FILE]] [--materials [--dump-materials FILE]] [--gpus N [--logical]]\n",
                     argv[0]);
        return 2;
    }
    const std::string path = argv[1];
    float vs = 0.f;
    try { vs = std::stof(argv[2]); } catch (const std::exception&) { std::fprintf(stderr, "invalid voxel size '%s'\n", argv[2]); return 2; }
    std::string grid = "bool", dumpFile, renderFile, matDump, cameraDump;
    uint32_t rw = 1280, rh = 720;  // main.cpp:72-73
    bool parallel = false, materials = false, logical = false;
    int gpus = 1;
    long benchRuns = 0;
    for (int i = 3; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--grid") && i + 1 < argc) grid = argv[++i];
        else if (!std::strcmp(argv[i], "--parallel")) parallel = true;
        else if (!std::strcmp(argv[i], "--dump") && i + 1 < argc) dumpFile = argv[++i];
        else if (!std::strcmp(argv[i], "--bench") && i + 1 < argc) benchRuns = std::atol(argv[++i]);
        else if (!std::strcmp(argv[i], "--render") && i + 1 < argc) renderFile = argv[++i];
        else if (!std::strcmp(argv[i], "--materials")) materials = true;
        else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--logical")) logical = true;
        else if (!std::strcmp(argv[i], "--dump-materials") && i + 1 < argc) matDump = argv[++i];
        else if (!std::strcmp(argv[i], "--camera-dump") && i + 1 < argc) cameraDump = argv[++i];
        else if (!std::strcmp(argv[i], "--size") && i + 1 < argc) { if (std::sscanf(argv[++i], "%ux%u", &rw, &rh) != 2) { std::fprintf(stderr, "bad --size\n"); return 2; } }
        else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    std::vector<int> devices;
    if (gpus > 1) {
        const int have = vx_device_count();
        if (!logical && gpus > have) { std::fprintf(stderr, "--gpus %d but %d device(s) visible (add --logical to rehearse on one)\n", gpus, have); return 2; }
        for (int k = 0; k < gpus; ++k) devices.push_back(logical ? 0 : k);
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
        if (grid == "bool") return parallel ? run_grid<VoxelGridBool, true>(path, vs, dumpFile, "VoxelGridBool", renderFile, rw, rh, materials, matDump, cameraDump, devices)
                                               : run_grid<VoxelGridBool, false>(path, vs, dumpFile, "VoxelGridBool", renderFile, rw, rh, materials, matDump, cameraDump, devices);
        if (grid == "aabbstruct") return parallel ? run_grid<VoxelGridAABBstruct, true>(path, vs, dumpFile, "VoxelGridAABBstruct") : run_grid<VoxelGridAABBstruct, false>(path, vs, dumpFile, "VoxelGridAABBstruct");
        if (grid == "vec") return parallel ? run_grid<VoxelGridVec, true>(path, vs, dumpFile, "VoxelGridVec") : run_grid<VoxelGridVec, false>(path, vs, dumpFile, "VoxelGridVec");
        std::fprintf(stderr, "unknown grid flavour %s\n", grid.c_str());
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
