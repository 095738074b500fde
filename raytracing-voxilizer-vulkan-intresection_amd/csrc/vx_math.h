// vx_math.h -- float32 arithmetic of the voxelizer / ray path, shared by host and gfx950 device code.
//
// The reference's occupancy is a float-op-order contract: every expression below keeps the
// association of the reference expression it implements, and the translation unit is compiled with
// -ffp-contract=off (plus the pragma below) so that no a*b+c is fused.  Division is IEEE-correct
// (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).  Inputs are assumed finite.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VX_HD __host__ __device__ __forceinline__
#else
#define VX_HD inline
#endif

#pragma clang fp contract(off)

namespace vx {

struct f3 { float x, y, z; };

struct GridParams {
    float org[3];     // m_org                                  voxelgrid.hpp:25
    float vs;         // m_voxelSize
    float half;       // vs * 0.5f  (VoxelBuilder.hpp:407) == 0.5f * vs (voxelgridBool.cpp:22)
    uint32_t dim[3];  // m_x, m_y, m_z
    uint64_t nvox;    // X*Y*Z
    uint64_t nwords;  // ceil(nvox/32)
};

// min/max of three: only ever compared or truncated afterwards, so the sign of a zero result is immaterial and
// v_min3_f32 / v_max3_f32 may be used (std::min/glm::min differ from fminf only in NaN and zero-sign handling).
VX_HD float min3(float a, float b, float c) { return fminf(a, fminf(b, c)); }
VX_HD float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }

// VoxelGrid::getCorrds, one axis: m_org + (pos + 0.5f) * m_voxelSize              voxelgrid.hpp:96-97
VX_HD float cell_centre(float org, float vs, uint32_t i) { return org + (((float)i + 0.5f) * vs); }

// The Aabb every grid flavour emits for voxel (x,y,z): c - half, c + half          voxelgridBool.cpp:37-41,
// voxelgridAABBstruct.cpp:35-42, voxelgridVecEncoding.cpp:30-36, octTree.hpp:237-240,382
VX_HD void cell_aabb(const GridParams& g, uint32_t x, uint32_t y, uint32_t z, float out[6])
{
    const float cx = cell_centre(g.org[0], g.vs, x), cy = cell_centre(g.org[1], g.vs, y), cz = cell_centre(g.org[2], g.vs, z);
    out[0] = cx - g.half; out[1] = cy - g.half; out[2] = cz - g.half;
    out[3] = cx + g.half; out[4] = cy + g.half; out[5] = cz + g.half;
}

// Candidate voxel range of a triangle along one axis                              VoxelBuilder.hpp:175-184
//   start = max(0, int((triMin - gridMin) / voxelSize)),  end = min(int(dim), int((triMax - gridMin) / voxelSize) + 2)
// vsize is halfVoxelSize.x * 2.0f in the serial driver (:173) and voxelSize in the threaded one (:500).
VX_HD void cand_axis(float a, float b, float c, float gmin, float vsize, uint32_t dim, int& s, int& e)
{
    const float tmn = min3(a, b, c), tmx = max3(a, b, c);
    const int s0 = (int)((tmn - gmin) / vsize);
    const int e0 = (int)((tmx - gmin) / vsize) + 2;
    s = s0 > 0 ? s0 : 0;
    e = e0 < (int)dim ? e0 : (int)dim;
}

// One SAT axis: separated iff min3(d) > R || max3(d) < -R                          VoxelBuilder.hpp:83-85, :258-262
VX_HD bool separated(float d0, float d1, float d2, float R) { return (min3(d0, d1, d2) > R) | (max3(d0, d1, d2) < -R); }

// Triangle/box overlap, both reference variants:
//   EPS=true  triBoxOverlap               VoxelBuilder.hpp:118-162 (octTree.hpp:440-484): axes with |L|_1 < 1e-8 and
//                                          normals with |n|_1 < 1e-8 are skipped
//   EPS=false triBoxOverlapSchwarzSeidel  VoxelBuilder.hpp:226-335: no skips
// The two variants' projections are negations of each other term by term (e.g. :137 Lx=(0,-e.z,e.y) gives
// d = (-(p.y*e.z)) + p.z*e.y while :268 computes (-(p.z*e.y)) + p.y*e.z); the test is symmetric in d -> -d, the
// radii are the same expression, so one body serves both.  The order of the 13 tests is free (pure conjunction), and they are
// combined with | and & on purpose: with || and && the device compiler puts a branch around every skippable axis (seven
// exec-mask regions per voxel, the verdicts materialised in VGPRs: 175 VALU instructions per voxel against 137 without).
//
// SatRow holds everything that does not depend on the voxel's x index (p.y, p.z, e.y, e.z and the three e x X axes),
// so a lane sweeping a row of voxels along x pays for it once.
struct SatRow {
    float p0y, p0z, p1y, p1z, p2y, p2z;
    float e0y, e0z, e1y, e1z, e2y, e2z;
    float nx;      // cross(e0,e1).x = e0.y*e1.z - e1.y*e0.z
    bool alive;    // false: the y/z slab tests or an e x X axis already separate every voxel of the row
};

template <bool EPS>
VX_HD SatRow sat_row_setup(const float v[9], float cy, float cz, float h)
{
    SatRow r;
    r.p0y = v[1] - cy; r.p0z = v[2] - cz;
    r.p1y = v[4] - cy; r.p1z = v[5] - cz;
    r.p2y = v[7] - cy; r.p2z = v[8] - cz;
    r.e0y = r.p1y - r.p0y; r.e0z = r.p1z - r.p0z;
    r.e1y = r.p2y - r.p1y; r.e1z = r.p2z - r.p1z;
    r.e2y = r.p0y - r.p2y; r.e2z = r.p0z - r.p2z;
    bool sep = false;
    // box axes y, z                                                           VoxelBuilder.hpp:94-100
    sep |= (min3(r.p0y, r.p1y, r.p2y) > h) | (max3(r.p0y, r.p1y, r.p2y) < -h);
    sep |= (min3(r.p0z, r.p1z, r.p2z) > h) | (max3(r.p0z, r.p1z, r.p2z) < -h);
    // e x X = (0, -e.z, e.y): R = h.y*|e.z| + h.z*|e.y|; d = (-(p.y*e.z)) + p.z*e.y      VoxelBuilder.hpp:137-139
#define VX_AXIS_X(ey, ez)                                                                   \
    {                                                                                        \
        const float R = h * fabsf(ez) + h * fabsf(ey);                                       \
        const bool live = !EPS || !((fabsf(ez) + fabsf(ey)) < 1e-8f);                        \
        const float d0 = (-(r.p0y * (ez))) + r.p0z * (ey);                                   \
        const float d1 = (-(r.p1y * (ez))) + r.p1z * (ey);                                   \
        const float d2 = (-(r.p2y * (ez))) + r.p2z * (ey);                                   \
        sep |= live & separated(d0, d1, d2, R);  /* (& not &&: no branch around six multiplies) */                                             \
    }
    VX_AXIS_X(r.e0y, r.e0z)
    VX_AXIS_X(r.e1y, r.e1z)
    VX_AXIS_X(r.e2y, r.e2z)
#undef VX_AXIS_X
    r.nx = r.e0y * r.e1z - r.e1y * r.e0z;
    r.alive = !sep;
    return r;
}

// BOXX = false leaves out the box-axis test along x.  Only for callers that sweep x over a range trimmed with that very test (K2a's
// trim_axis: the same cell_centre, the same v - c, the same min3 / max3 against half): float subtraction and cell_centre are monotone, so
// the test separates on a prefix and a suffix of the x range only, and the trimmed range has neither.
// LIVE = false leaves out the EPS variant's "axis too short" checks.  Only for callers that know every one of them to be false from the
// row's own constants: |e.z| + |e.x| >= |e.z| in float, so an axis with |e.z| >= 1e-8 (e x Y), |e.y| >= 1e-8 (e x Z) or |n.x| >= 1e-8
// (the normal) is tested whatever x is (sat_row_all_live).
template <bool EPS, bool BOXX = true, bool LIVE = true>
VX_HD bool sat_row_test(const SatRow& r, const float v[9], float cx, float h)
{
    const float p0x = v[0] - cx, p1x = v[3] - cx, p2x = v[6] - cx;
    const float e0x = p1x - p0x, e1x = p2x - p1x, e2x = p0x - p2x;
    bool sep = BOXX ? ((min3(p0x, p1x, p2x) > h) | (max3(p0x, p1x, p2x) < -h)) : false;   // box axis x   :90-92
    // e x Y = (e.z, 0, -e.x): R = h.x*|e.z| + h.z*|e.x|; d = p.x*e.z + (-(p.z*e.x))          :141-143
    // e x Z = (-e.y, e.x, 0): R = h.x*|e.y| + h.y*|e.x|; d = (-(p.x*e.y)) + p.y*e.x          :145-147
#define VX_AXIS_YZ(ex, ey, ez)                                                               \
    {                                                                                        \
        const float Ry = h * fabsf(ez) + h * fabsf(ex);                                      \
        const bool livey = !EPS || !LIVE || !((fabsf(ez) + fabsf(ex)) < 1e-8f);             \
        const float a0 = p0x * (ez) + (-(r.p0z * (ex)));                                     \
        const float a1 = p1x * (ez) + (-(r.p1z * (ex)));                                     \
        const float a2 = p2x * (ez) + (-(r.p2z * (ex)));                                     \
        sep |= livey & separated(a0, a1, a2, Ry);                                           \
        const float Rz = h * fabsf(ey) + h * fabsf(ex);                                      \
        const bool livez = !EPS || !LIVE || !((fabsf(ey) + fabsf(ex)) < 1e-8f);             \
        const float b0 = (-(p0x * (ey))) + r.p0y * (ex);                                     \
        const float b1 = (-(p1x * (ey))) + r.p1y * (ex);                                     \
        const float b2 = (-(p2x * (ey))) + r.p2y * (ex);                                     \
        sep |= livez & separated(b0, b1, b2, Rz);                                           \
    }
    VX_AXIS_YZ(e0x, r.e0y, r.e0z)
    VX_AXIS_YZ(e1x, r.e1y, r.e1z)
    VX_AXIS_YZ(e2x, r.e2y, r.e2z)
#undef VX_AXIS_YZ
    // plane: n = cross(e0,e1); r = (h*|n.x| + h*|n.y|) + h*|n.z|; s = (n.x*p0.x + n.y*p0.y) + n.z*p0.z   :104-115,157-158
    const float ny = r.e0z * e1x - r.e1z * e0x;
    const float nz = e0x * r.e1y - e1x * r.e0y;
    const float anx = fabsf(r.nx), any = fabsf(ny), anz = fabsf(nz);
    const bool livep = !EPS || !LIVE || !(((anx + any) + anz) < 1e-8f);
    const float rr = (h * anx + h * any) + h * anz;
    const float s = (r.nx * p0x + ny * r.p0y) + nz * r.p0z;
    sep |= livep & (fabsf(s) > rr);
    return !sep;
}

// True when none of the EPS variant's x-dependent "axis too short" checks can fire on this row (see sat_row_test, LIVE)
VX_HD bool sat_row_all_live(const SatRow& r)
{
    const float t = 1e-8f;
    return (fabsf(r.e0y) >= t) & (fabsf(r.e0z) >= t) & (fabsf(r.e1y) >= t) & (fabsf(r.e1z) >= t) & (fabsf(r.e2y) >= t) & (fabsf(r.e2z) >= t) & (fabsf(r.nx) >= t);
}

// Whole test for one voxel (used by host-side checks and single-voxel paths)
template <bool EPS>
VX_HD bool tri_box_overlap(const float v[9], float cx, float cy, float cz, float h)
{
    const SatRow r = sat_row_setup<EPS>(v, cy, cz, h);
    return r.alive && sat_row_test<EPS>(r, v, cx, h);
}

// hitAabb                                                                      shaders/raytrace.rint:46-56
//   invDir = 1.0/dir; tbot = invDir*(min - o); ttop = invDir*(max - o); t0 = max3(min(ttop,tbot)); t1 = min3(max(ttop,tbot))
//   return t1 > max(t0, 0.0) ? t0 : -1.0
VX_HD float hit_aabb(const float b[6], const float o[3], const float inv[3])
{
    const float bx = inv[0] * (b[0] - o[0]), tx = inv[0] * (b[3] - o[0]);
    const float by = inv[1] * (b[1] - o[1]), ty = inv[1] * (b[4] - o[1]);
    const float bz = inv[2] * (b[2] - o[2]), tz = inv[2] * (b[5] - o[2]);
    const float t0 = fmaxf(fminf(tx, bx), fmaxf(fminf(ty, by), fminf(tz, bz)));
    const float t1 = fminf(fmaxf(tx, bx), fminf(fmaxf(ty, by), fmaxf(tz, bz)));
    return t1 > fmaxf(t0, 0.0f) ? t0 : -1.0f;
}

// Octree::morton3D                                                              octTree.hpp:211-218
// The reference composes three byte-LUT lookups with shifts <<48 then <<24, so the contribution of coordinate
// bits 16..23 leaves the 64-bit word: the result is the plain interleave of the low 16 bits of x, y, z.
VX_HD uint64_t part1by2_16(uint32_t v)
{
    uint64_t x = v & 0xFFFFu;                       // 16 bits -> every third bit of 48
    x = (x | (x << 16)) & 0x0000FF0000FFull;        // ........ 8 ........ 8
    x = (x | (x << 8)) & 0x00F00F00F00Full;
    x = (x | (x << 4)) & 0x0C30C30C30C3ull;
    x = (x | (x << 2)) & 0x249249249249ull;
    return x;
}
VX_HD uint64_t morton3d(uint32_t x, uint32_t y, uint32_t z) { return part1by2_16(x) | (part1by2_16(y) << 1) | (part1by2_16(z) << 2); }

// Octree::compactBits / decodeMortonToVoxel                                     octTree.hpp:220-236
VX_HD uint32_t compact_bits(uint64_t v)
{
    v &= 0x1249249249249249ull;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ull;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00full;
    v = (v ^ (v >> 8)) & 0x1f0000ff0000ffull;
    v = (v ^ (v >> 16)) & 0x1f00000000ffffull;
    v = (v ^ (v >> 32)) & 0x1fffffull;
    return (uint32_t)v;
}

}  // namespace vx
