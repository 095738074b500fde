// vx_walk.hip -- K6: first hit per ray against the occupied voxels' AABBs (replaces the procedural-hit stage raytrace.rint:46-71
// that the reference runs under traceRayEXT, raytrace.rgen:49-64) -- major-axis slab walk.
//
// The reference hands the occupied voxels' AABBs to the driver's BVH and runs raytrace.rint on every candidate; the result per
// ray is the minimum over ALL boxes of t0 = hitAabb(box) subject to t0 > 0 (rint:69) and tmin <= t0 <= tmax.  Here the occupancy
// itself is the acceleration structure, and the enumeration of candidate cells is organised so that it is cheap, uniform across
// the lanes of a wave, and provably a superset of the cells whose float boxes the ray can enter:
//
//   Let w be the axis of the largest |direction| component, u and v the other two.  The grid is cut into slabs perpendicular to
//   w at three granularities: 64 cells (block slabs, level 2), 8 cells (brick slabs, level 1), 1 cell (level 0).  For a slab with lattice
//   planes P_near, P_far (in travel order) the ray is within the position tolerance `tol` of the slab for
//        t in [ta, tb],  ta = inv_w * ((P_near -/+ tol) - o_w),  tb = inv_w * ((P_far +/- tol) - o_w)
//   -- the same expression form as hitAabb's `invDir * (plane - origin)` (rint:49-50), so by monotonicity of float subtraction
//   and multiplication every box of the slab has a COMPUTED entry time >= ta: once the best accepted t is < ta of a slab, no box
//   of that slab or of any later one can beat it (exact, no slack needed).  Inside [ta, tb] the ray's u and v coordinates stay in
//        [min(p(ta), p(tb)) - 2 tol, max(p(ta), p(tb)) + 2 tol]
//   (|d_u|, |d_v| <= |d_w|: an error in t moves the point by less than the same error along w; no 1/d_u blow-up exists, and a
//   zero component simply gives a constant coordinate), so the cells the ray can touch in the slab lie in that rectangle --
//   1x1 .. 2x2 cells of the slab's granularity, 3x3 at worst.
//
// A step of the walk = one slab, the same code at every level: the slab's [ta, tb], the termination test, the rectangle.  A
// block slab (level 2) or brick slab (level 1) looks its rectangle up in the occupancy mip of its level; an occupied block
// rectangle descends to its eight brick slabs; the occupied bricks of a brick slab's rectangle are walked one after the other at
// level 0.  Entering a brick costs ONE 64-byte line -- its eight slab words -- and drops the slabs whose word misses the
// rectangle the ray sweeps across the whole brick; each remaining slab ANDs its own (tighter) rectangle with its word, and the
// surviving cells go through the exact rint formula on the float box the reference would have built for them, which is the only
// arbiter of hit and t -- the reported t is the very float the brute-force minimum yields.
//
// Data layout (built once per bitmask by k_build_bricks3 / k_brick_mip1 / k_build_mip2, the analogue of the reference's BLAS
// build, hello_vulkan.cpp:737-760):
//   level 0  "bricks": the bitmask re-tiled brick-major in THREE orientations, one per possible major axis: for orientation w one
//            uint64 per (8x8x8 brick, slab along w), bit = (v&7)*8 + (u&7) with (u, v) = the axes after w cyclically.  Whatever
//            the ray's major axis, a brick is one contiguous 64-byte line of eight slab words.
//   level 1  one bit per brick, x-fastest; level 2 one bit per 8^3 bricks; both staged in LDS by every workgroup when they fit.
//
// Wave efficiency.  Persistent waves; a lane whose ray has finished takes the next ray of its wave's chunk of a global queue
// (first chunk static, later ones by one atomicAdd of a guided size).  Every busy lane executes the same slab step whatever its
// level; only the mip lookups, the brick fetch and the exact tests (1.7 per ray on the bench scene) diverge.  Once the queue is
// dry, lanes with a long interval left hand its far half to idle lanes of their wave; the pieces meet in LDS.  Measured history
// and lane-utilisation figures: DESIGN.md section 4.
//
// No MFMA: this is traversal, not a contraction.  Algorithmic HBM traffic is the ray stream (24 B in, 4-8 B out per ray).
#include "vx_internal.h"

#include <cstddef>
#include <cstdlib>
#include <cstring>

#pragma clang fp contract(off)

namespace vx {

#define VX_KL(kern, grid, block, shmem, stream, ...)                         \
    do {                                                                     \
        ProfScope ps_(#kern, stream);                                        \
        hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);   \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// Brick-major re-tiling of the occupancy bitmask through LDS, three orientations.  A workgroup takes 64 bricks in a row along
// x for one (by, bz): 64 voxel rows (8 z x 8 y) of 512 voxels.  Reads: row-contiguous words, funnel-shifted to the chunk's own
// 32-voxel alignment.  The z orientation (bit y*8+x per z slab) falls out of the staged rows directly; the x orientation (bit
// z*8+y per x slab) and the y orientation (bit x*8+z per y slab) are 8x8 bit transposes of it.  Writes: per orientation the 64
// bricks' 8 words each = 4 KiB contiguous per workgroup.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long transpose8x8(unsigned long long x)  // byte i bit j -> byte j bit i
{
    x = (x & 0xAA55AA55AA55AA55ull) | ((x & 0x00AA00AA00AA00AAull) << 7) | ((x >> 7) & 0x00AA00AA00AA00AAull);
    x = (x & 0xCCCC3333CCCC3333ull) | ((x & 0x0000CCCC0000CCCCull) << 14) | ((x >> 14) & 0x0000CCCC0000CCCCull);
    x = (x & 0xF0F0F0F00F0F0F0Full) | ((x & 0x00000000F0F0F0F0ull) << 28) | ((x >> 28) & 0x00000000F0F0F0F0ull);
    return x;
}

// `m1` (optional; needs BX % 64 == 0 so that a workgroup's 64 bricks are two whole words of it): the level-1 mip, one bit per brick,
// written from here -- and then EMPTY bricks are not stored at all (the walk never fetches a brick whose level-1 bit is clear):
// on a surface scene that is three quarters of the 3 x N/8 bytes this kernel would write.
__global__ __launch_bounds__(256) void k_build_bricks3(uint32_t* __restrict__ words /*read; written when `tiled` is given*/, uint32_t X, uint32_t Y, uint32_t Z, uint32_t BX, uint32_t BY,
                                                       uint32_t BZ, uint32_t chunks_x, uint64_t nwords, unsigned long long* __restrict__ bricks3,
                                                       uint64_t ori_stride /*uint64 words per orientation*/, uint32_t* __restrict__ m1,
                                                       const uint32_t* __restrict__ tiled /*optional (X % 32 == 0): the voxelizer's tiled build mask is the
                                                       source, and `words` -- the reference's bitmask -- is WRITTEN from it on the way (k_untile's job)*/)
{
    __shared__ uint32_t rows[64][17];              // [z*8 + y][32-voxel chunk of the 512]; padded against bank conflicts of the column reads
    __shared__ unsigned long long sz[64][9];       // [brick][z slab]: bit y*8 + x (padded)
    const uint64_t ngroups = (uint64_t)chunks_x * BY * BZ;
    for (uint64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint32_t cx = (uint32_t)(grp % chunks_x);
        const uint64_t q = grp / chunks_x;
        const uint32_t by = (uint32_t)(q % BY), bz = (uint32_t)(q / BY);
        const uint32_t x0 = cx * 512u;
        // ---- load: 64 rows x 16 words.  Rows of a multiple of 128 voxels start on 16-byte boundaries: one 16-byte load per thread
        // (four words of one row) instead of four 4-byte loads with funnel shifts -- the kernel is bound by the latency of these
        // strided loads, not by their bytes.
        uint32_t nonzero = 0u;
        if (tiled) {
            // the group's 64 rows are 2 x 2 tile rows (y / 4, z / 4) of the tiled mask, 16 tiles of 64 bytes each: one 16-byte load per
            // thread (the four y words of one z of a tile)
            const uint32_t xw = X >> 5, tiles_y = (Y + 3u) >> 2, tiles_z = (Z + 3u) >> 2;
            const uint32_t q = threadIdx.x >> 6, l = threadIdx.x & 63u;
            const uint32_t ty = by * 2u + (q & 1u), tz = bz * 2u + (q >> 1), xt = l >> 2, zz = l & 3u, xs = cx * 16u + xt;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (ty < tiles_y && tz < tiles_z && xs < xw) val = *reinterpret_cast<const uint4*>(tiled + (((uint64_t)tz * tiles_y + ty) * xw + xs) * 16ull + zz * 4u);
            const uint32_t r0 = ((q >> 1) * 4u + zz) * 8u + (q & 1u) * 4u;  // row (z % 8) * 8 + y % 8 of the first of the four words
            rows[r0][xt] = val.x; rows[r0 + 1u][xt] = val.y; rows[r0 + 2u][xt] = val.z; rows[r0 + 3u][xt] = val.w;
            nonzero = val.x | val.y | val.z | val.w;
        } else if ((X & 127u) == 0u) {
            const uint32_t r = threadIdx.x >> 2, q4 = (threadIdx.x & 3u) * 4u;
            const uint32_t z = bz * 8u + (r >> 3), y = by * 8u + (r & 7u), xs = x0 + q4 * 32u;
            uint4 val = make_uint4(0u, 0u, 0u, 0u);
            if (z < Z && y < Y && xs < X) val = *reinterpret_cast<const uint4*>(words + (((uint64_t)X * ((uint64_t)y + (uint64_t)Y * z) + xs) >> 5));
            rows[r][q4] = val.x; rows[r][q4 + 1u] = val.y; rows[r][q4 + 2u] = val.z; rows[r][q4 + 3u] = val.w;
            nonzero = val.x | val.y | val.z | val.w;
        } else
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t item = (uint32_t)k * 256u + threadIdx.x;
            const uint32_t r = item >> 4, j = item & 15u;
            const uint32_t z = bz * 8u + (r >> 3), y = by * 8u + (r & 7u), xs = x0 + j * 32u;
            uint32_t val = 0u;
            if (z < Z && y < Y && xs < X) {
                const uint64_t i0 = (uint64_t)X * ((uint64_t)y + (uint64_t)Y * z) + xs;
                const uint32_t sh = (uint32_t)i0 & 31u;
                const uint64_t wi = i0 >> 5;
                val = words[wi] >> sh;
                if (sh && wi + 1 < nwords) val |= words[wi + 1] << (32u - sh);
                const uint32_t nb = X - xs;  // voxels of this row left from xs on
                if (nb < 32u) val &= (1u << nb) - 1u;
            }
            rows[r][j] = val;
            nonzero |= val;
        }
        bool empty = false;
        if (m1) {
            // an empty row of 64 bricks (most of them, on a surface scene): two zero words of the mip, nothing else
            empty = !__syncthreads_or(nonzero != 0u);
            if (empty && threadIdx.x < 2u) m1[((uint64_t)cx * 64u + (uint64_t)BX * ((uint64_t)by + (uint64_t)BY * bz)) / 32u + threadIdx.x] = 0u;
        } else {
            __syncthreads();
        }
        if (tiled) {
            // the reference's bitmask, written from the staged rows: four consecutive words of a row per thread
            const uint32_t xw = X >> 5;
            const uint32_t r = threadIdx.x >> 2, q4 = (threadIdx.x & 3u) * 4u;
            const uint32_t z = bz * 8u + (r >> 3), y = by * 8u + (r & 7u), xs = cx * 16u + q4;
            if (z < Z && y < Y && xs < xw) {
                uint32_t* dst = words + (uint64_t)xw * ((uint64_t)y + (uint64_t)Y * z) + xs;
                if (empty) {
                    if ((xw & 3u) == 0u) *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
                    else for (uint32_t k = 0; k < 4u && xs + k < xw; ++k) dst[k] = 0u;
                } else if ((xw & 3u) == 0u) {
                    *reinterpret_cast<uint4*>(dst) = make_uint4(rows[r][q4], rows[r][q4 + 1u], rows[r][q4 + 2u], rows[r][q4 + 3u]);
                } else {
                    for (uint32_t k = 0; k < 4u && xs + k < xw; ++k) dst[k] = rows[r][q4 + k];
                }
            }
        }
        if (empty) continue;  // (nothing of `rows` is read on this path: no barrier needed before the next group's loads overwrite it)
        // ---- z orientation (bit y*8 + x per z slab): two (brick, slab) pairs per thread, into LDS
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t item = (uint32_t)k * 256u + threadIdx.x;
            const uint32_t b = item >> 3, sl = item & 7u;
            unsigned long long bits = 0ull;
            const uint32_t sh = (b & 3u) * 8u;
#pragma unroll
            for (uint32_t yy = 0; yy < 8u; ++yy) bits |= (unsigned long long)((rows[sl * 8u + yy][b >> 2] >> sh) & 0xFFu) << (yy * 8u);
            sz[b][sl] = bits;
        }
        __syncthreads();
        // ---- stores: thread = (brick, part); part 0,1: x slabs 0-3 / 4-7, part 2,3: y slabs 0-3 / 4-7, and every part two of the z slabs
        {
            const uint32_t b = threadIdx.x & 63u, part = threadIdx.x >> 6;
            const uint32_t bx = cx * 64u + b;
            unsigned long long s[8];
            unsigned long long any = 0ull;
#pragma unroll
            for (int z = 0; z < 8; ++z) { s[z] = sz[b][z]; any |= s[z]; }
            const uint64_t brick = (uint64_t)bx + (uint64_t)BX * ((uint64_t)by + (uint64_t)BY * bz);
            if (m1 && part == 0u) {
                const unsigned long long occ = __ballot(any != 0ull);  // BX % 64 == 0: all 64 bricks exist, `brick` of lane 0 is a multiple of 64
                if (b == 0u) {
                    m1[brick >> 5] = (uint32_t)occ;
                    m1[(brick >> 5) + 1u] = (uint32_t)(occ >> 32);
                }
            }
            if (bx < BX && (any || !m1)) {
                const uint32_t ori = part >> 1, s0 = (part & 1u) * 4u;
                unsigned long long* dst = bricks3 + (uint64_t)ori * ori_stride + brick * 8ull + s0;
                unsigned long long out[4] = {0ull, 0ull, 0ull, 0ull};
                if (any) {
                    if (ori == 0) {
                        // X[x]: byte z = (transpose of slab z).byte x  -> bit z*8 + y
#pragma unroll
                        for (int z = 0; z < 8; ++z) {
                            const unsigned long long t = transpose8x8(s[z]);  // byte x, bit y
#pragma unroll
                            for (int j = 0; j < 4; ++j) out[j] |= ((t >> (8u * (s0 + j))) & 0xFFull) << (8 * z);
                        }
                    } else {
                        // Y[y]: gather row y of every z slab (byte z = bits x), transpose -> byte x, bit z
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            unsigned long long gth = 0ull;
#pragma unroll
                            for (int z = 0; z < 8; ++z) gth |= ((s[z] >> (8u * (s0 + j))) & 0xFFull) << (8 * z);
                            out[j] = transpose8x8(gth);
                        }
                    }
                }
                reinterpret_cast<ulonglong2*>(dst)[0] = make_ulonglong2(out[0], out[1]);
                reinterpret_cast<ulonglong2*>(dst)[1] = make_ulonglong2(out[2], out[3]);
                unsigned long long* dz = bricks3 + 2ull * ori_stride + brick * 8ull + 2u * part;
                const unsigned long long z0 = part == 0u ? s[0] : (part == 1u ? s[2] : (part == 2u ? s[4] : s[6]));
                const unsigned long long z1 = part == 0u ? s[1] : (part == 1u ? s[3] : (part == 2u ? s[5] : s[7]));
                *reinterpret_cast<ulonglong2*>(dz) = make_ulonglong2(z0, z1);
            }
        }
        __syncthreads();
    }
}

// Returns true when the level-1 mip was written by the brick kernel itself (and empty bricks were left unwritten).
bool launch_build_bricks3(const uint32_t* words, const uint32_t dim[3], const uint32_t bdim[3], unsigned long long* bricks3, uint32_t* m1, hipStream_t s,
                          const uint32_t* tiled)
{
    const uint64_t n = (uint64_t)bdim[0] * bdim[1] * bdim[2];
    if (!n) return false;
    const uint32_t chunks_x = (bdim[0] + 63u) / 64u;
    const uint64_t ngroups = (uint64_t)chunks_x * bdim[1] * bdim[2];
    const uint64_t nvox = (uint64_t)dim[0] * dim[1] * dim[2];
    const uint64_t nwords = (nvox + 31) / 32;
    uint64_t nblk = ngroups;
    if (nblk > 16384) nblk = 16384;
    const bool fused = (bdim[0] % 64u) == 0u && m1 != nullptr;
    VX_KL(k_build_bricks3, dim3((unsigned)nblk), dim3(256), 0, s, const_cast<uint32_t*>(words), dim[0], dim[1], dim[2], bdim[0], bdim[1], bdim[2], chunks_x, nwords, bricks3, n * 8ull,
          fused ? m1 : nullptr, (dim[0] % 32u) == 0u ? tiled : nullptr);
    return fused;
}

namespace {

// Everything a lane carries for the ray it is currently tracing, in PERMUTED axis order: w = the major axis, u and v the axes
// after it cyclically.  hitAabb (rint:46-56) is a max of per-axis minima and a min of per-axis maxima, so evaluating it in
// (u, v, w) order yields the same float as in (x, y, z) order; only the voxel index of a hit needs the real order.
// Two cell (or brick) coordinates in one register: 16 + 16 bits -- every grid whose axes have at most 65535 cells -- or, WIDE, 32 + 32
// bits (grids with an axis of up to 2^21 cells; the reference's dims are size_t, VoxelBuilder.hpp:347-349).  The wide form costs four
// registers and is compiled only into the variants that walk such grids.
template <bool WIDE> struct Pk { typedef uint32_t T; };
template <> struct Pk<true> { typedef unsigned long long T; };
template <bool WIDE> __device__ __forceinline__ typename Pk<WIDE>::T pk_make(int lo, int hi)
{
    typedef typename Pk<WIDE>::T T;
    return WIDE ? (T)((T)(uint32_t)lo | ((T)(uint32_t)hi << (WIDE ? 32 : 16))) : (T)((uint32_t)lo | ((uint32_t)hi << 16));
}
template <bool WIDE> __device__ __forceinline__ int pk_lo(typename Pk<WIDE>::T v) { return WIDE ? (int)(uint32_t)v : (int)((uint32_t)v & 0xFFFFu); }
template <bool WIDE> __device__ __forceinline__ int pk_hi(typename Pk<WIDE>::T v) { return WIDE ? (int)(uint32_t)((unsigned long long)v >> 32) : (int)((uint32_t)v >> 16); }

template <typename IdxT, bool WIDE = false>
struct WalkLane {
    float ou, ov, ow;        // origin
    float du, dv;            // direction (the major component is only needed through its reciprocal)
    float iu, iv, iw;        // 1 / direction (rint:48)
    float orgu, orgv, orgw;  // grid origin
    int dimu, dimv, dimw;    // grid dims in cells
    int perm;                // w: 0 x, 1 y, 2 z
    float tol;               // position tolerance
    float tn, tf;            // the ray inside the dilated grid box, cut to [0, tmax]
    float tmax;
    float best;              // best accepted t so far (+inf: none)
    IdxT best_idx;           // voxel index of the best hit (all ones: none)
    int lvl, k;              // current slab: level (2 block slabs, 1 brick slabs; 0: inside a brick of brick slab k) and its index along w
    int cw0;                 // cell index along w where this ray (or piece) starts, one cell of slack: slabs wholly in front of it are never looked at
    // occupied bricks of the current brick slab's rectangle, waiting for the brick phase
    uint32_t pend;           // bits 0..15: brick jb + j = (cu0 + (jb + j) % nu, cv0 + (jb + j) / nu) is occupied; bits 16..30: nu (WIDE: in nu_w); bit 31: more windows
    int jb;                  // first rectangle cell of the current 16-cell window (0 unless the rectangle has more than 16 cells)
    typename Pk<WIDE>::T pc;      // cu0 | cv0 << 16 (WIDE: << 32)
    typename Pk<WIDE>::T pu, pv;  // the slab's cell rectangle: u0 | u1 << 16, v0 | v1 << 16
    int nu_w;                // WIDE only: the rectangle's width in bricks (does not fit 15 bits)
    // level 0: the brick being walked -- its eight slab words (one 64-byte line) and the slabs still to look at
    unsigned long long w0, w1, w2, w3, w4, w5, w6, w7;
    uint32_t sm;             // bit s: slab s of the brick may hold a cell the ray touches (word & brick-level rectangle != 0)
};

// the low byte of m in all eight bytes (two shift-ors; a 64-bit multiply by 0x0101010101010101 is three quarter-rate instructions)
__device__ __forceinline__ unsigned long long rep8(uint32_t m)
{
    uint32_t r = m | (m << 8);
    r |= r << 16;
    return ((unsigned long long)r << 32) | r;
}
__device__ __forceinline__ float sel3f(int p, float a, float b, float c) { return p == 0 ? a : (p == 1 ? b : c); }
__device__ __forceinline__ unsigned long long sel8(int i, unsigned long long a, unsigned long long b, unsigned long long c, unsigned long long d,
                                                   unsigned long long e, unsigned long long f, unsigned long long g, unsigned long long h)
{
    const unsigned long long lo = i & 2 ? (i & 1 ? d : c) : (i & 1 ? b : a);
    const unsigned long long hi = i & 2 ? (i & 1 ? h : g) : (i & 1 ? f : e);
    return i & 4 ? hi : lo;
}
__device__ __forceinline__ int sel3i(int p, int a, int b, int c) { return p == 0 ? a : (p == 1 ? b : c); }

// (cu, cv, cw) in permuted order -> (x, y, z):  w=0: x=cw y=cu z=cv;  w=1: x=cv y=cw z=cu;  w=2: x=cu y=cv z=cw
__device__ __forceinline__ void unperm(int p, int cu, int cv, int cw, int& x, int& y, int& z)
{
    x = sel3i(p, cw, cv, cu);
    y = sel3i(p, cu, cw, cv);
    z = sel3i(p, cv, cu, cw);
}

// Ray r of the batch: from the ray buffer, or generated from the reference camera model (raytrace.rgen:41-47; mat*vec in glm's
// association (m0*v0 + m1*v1) + (m2*v2 + m3*v3)).
__device__ __forceinline__ void load_ray_w(bool primary, uint64_t r, const float* __restrict__ rays, const Camera* __restrict__ camp, float& ox, float& oy,
                                           float& oz, float& dx, float& dy, float& dz)
{
    if (primary) {
        const Camera& cam = *camp;
        const uint32_t px = (uint32_t)(r % cam.width), py = (uint32_t)(r / cam.width);
        const float u = ((float)px + 0.5f) / (float)cam.width, v = ((float)py + 0.5f) / (float)cam.height;
        const float ndx = u * 2.0f - 1.0f, ndy = v * 2.0f - 1.0f;
        float tg[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            tg[k] = (cam.projInv[0 + k] * ndx + cam.projInv[4 + k] * ndy) + (cam.projInv[8 + k] * 1.0f + cam.projInv[12 + k] * 1.0f);
        const float il = 1.0f / sqrtf((tg[0] * tg[0] + tg[1] * tg[1]) + tg[2] * tg[2]);
        const float n0 = tg[0] * il, n1 = tg[1] * il, n2 = tg[2] * il;
        ox = cam.viewInv[12]; oy = cam.viewInv[13]; oz = cam.viewInv[14];
        dx = (cam.viewInv[0] * n0 + cam.viewInv[4] * n1) + cam.viewInv[8] * n2;
        dy = (cam.viewInv[1] * n0 + cam.viewInv[5] * n1) + cam.viewInv[9] * n2;
        dz = (cam.viewInv[2] * n0 + cam.viewInv[6] * n1) + cam.viewInv[10] * n2;
    } else {
        const float2* rp = reinterpret_cast<const float2*>(rays + 6 * r);
        const float2 a = rp[0], b = rp[1], c = rp[2];
        ox = a.x; oy = a.y; oz = b.x; dx = b.y; dy = c.x; dz = c.y;
    }
}

// Ray set-up: major axis, tolerance, grid clip, first slab.  Returns false when the ray cannot touch the grid.
template <typename IdxT, bool WIDE>
__device__ __forceinline__ bool walk_setup(WalkLane<IdxT, WIDE>& R, const GridParams& g, float inv_vs, float tmax, float ox, float oy, float oz, float dx, float dy,
                                           float dz)
{
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;  // rint:48
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    const int p = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    R.perm = p;
    R.ow = sel3f(p, ox, oy, oz); R.iw = sel3f(p, ix, iy, iz);
    R.ou = sel3f(p, oy, oz, ox); R.du = sel3f(p, dy, dz, dx); R.iu = sel3f(p, iy, iz, ix);
    R.ov = sel3f(p, oz, ox, oy); R.dv = sel3f(p, dz, dx, dy); R.iv = sel3f(p, iz, ix, iy);
    R.orgw = sel3f(p, g.org[0], g.org[1], g.org[2]);
    R.orgu = sel3f(p, g.org[1], g.org[2], g.org[0]);
    R.orgv = sel3f(p, g.org[2], g.org[0], g.org[1]);
    R.dimw = sel3i(p, (int)g.dim[0], (int)g.dim[1], (int)g.dim[2]);
    R.dimu = sel3i(p, (int)g.dim[1], (int)g.dim[2], (int)g.dim[0]);
    R.dimv = sel3i(p, (int)g.dim[2], (int)g.dim[0], (int)g.dim[1]);
    const float hx = g.org[0] + (float)g.dim[0] * g.vs, hy = g.org[1] + (float)g.dim[1] * g.vs, hz = g.org[2] + (float)g.dim[2] * g.vs;
    float Mx = fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz));
    Mx = fmaxf(Mx, fmaxf(fmaxf(fabsf(g.org[0]), fabsf(g.org[1])), fabsf(g.org[2])));
    Mx = fmaxf(Mx, fmaxf(fmaxf(fabsf(hx), fabsf(hy)), fabsf(hz)));
    // position tolerance 16 * 2^-24 * max|coordinate|: the box planes carry <= 3 roundings of grid-sized numbers, the slab
    // formula subtracts the (possibly far) origin and multiplies by a rounded reciprocal (relative 2^-23 of the distance
    // travelled), and this kernel's own plane times carry the same again
    const float tol = Mx * 9.5367431640625e-07f;
    R.tol = tol;
    float tn = 0.0f, tf = tmax;
    bool miss = !(ax > 0.0f || ay > 0.0f || az > 0.0f) || !g.nvox;  // a zero direction never reports a hit (inf/NaN slabs)
#define VX_CLIP(o, d, inv, lo, hi)                                                                   \
    {                                                                                                 \
        const float t1 = (((lo)-tol) - (o)) * (inv), t2 = (((hi) + tol) - (o)) * (inv);               \
        const bool z = (d) == 0.0f;                                                                   \
        miss |= z && (((o) < (lo)-tol) || ((o) > (hi) + tol));                                        \
        tn = fmaxf(tn, z ? -INFINITY : fminf(t1, t2));                                                \
        tf = fminf(tf, z ? INFINITY : fmaxf(t1, t2));                                                 \
    }
    VX_CLIP(ox, dx, ix, g.org[0], hx)
    VX_CLIP(oy, dy, iy, g.org[1], hy)
    VX_CLIP(oz, dz, iz, g.org[2], hz)
#undef VX_CLIP
    // the clip's own rounding: widen by the time it takes to travel 2 tol along the major axis
    const float tslack = 2.0f * tol * fabsf(R.iw);
    tn = fmaxf(tn - tslack, 0.0f);
    tf = tf + tslack;
    R.tn = tn;
    R.tf = tf;
    R.tmax = tmax;
    R.best = INFINITY;
    R.best_idx = (IdxT)~(IdxT)0;
    R.lvl = 2;
    R.k = 0;
    R.pend = 0u;
    R.jb = 0;
    R.pc = R.pu = R.pv = 0u;
    R.nu_w = 0;
    R.sm = 0u;
    R.w0 = R.w1 = R.w2 = R.w3 = R.w4 = R.w5 = R.w6 = R.w7 = 0ull;
    if (miss || !(tn <= tf)) return false;
    // first block slab: the one that holds the entry point (one cell of slack against the rounding of this estimate; slabs that
    // still lie in front of tn are skipped by the step itself)
    const bool pos = R.iw > 0.0f;
    const float pw = sel3f(p, ox + tn * dx, oy + tn * dy, oz + tn * dz);
    int cw = (int)floorf((pw - R.orgw) * inv_vs) + (pos ? -1 : 1);
    cw = cw < 0 ? 0 : (cw > R.dimw - 1 ? R.dimw - 1 : cw);
    R.k = cw >> 6;
    R.cw0 = cw;
    return true;
}

}  // namespace

// Everything the kernel is given, as ONE by-value argument.  The walk needs the hot block in scalar registers at every step; the
// cold block (ray source, outputs, work counter) is touched once per refill / retire.  Held as ordinary kernel arguments the
// cold values stay live in SGPRs for the whole kernel (and the camera's 34 floats get hoisted out of the loop on top), the SGPRs
// spill to VGPR lanes and the VGPRs to scratch; so the cold block is read from the kernarg segment where it is used, through a
// laundered pointer that the compiler cannot hoist loads from.
struct WalkHot {
    GridParams g;
    const unsigned long long* bricks3;  // [3 orientations][bricks][8 slabs]
    uint64_t ori_stride;                // uint64 words per orientation
    const uint32_t* w1;                 // level-1 mip: one bit per brick
    const uint32_t* w2;                 // level-2 mip: one bit per 8^3 bricks
    uint32_t d1[3], d2[3];
    float inv_vs;
    float tmin;
    int any_hit;
    uint32_t m1_words, m2_words;
    uint32_t nrays;               // one launch handles at most 2^31 rays (the host loops over larger batches)
    int donate;                   // intra-wave work donation in the drain phase (VOXHIP_TRACE_DONATE=0 switches it off)
};
struct WalkCold {
    const float* rays;            // null: primary rays from *cam
    const Camera* cam;
    const float* tmax_per_ray;    // null: tmax
    float tmax;
    uint32_t pad;
    uint64_t ray_base;            // index of this launch's first ray in the caller's batch (primary rays: pixel index)
    float* t_out;
    void* idx_out;                // IdxT per ray: voxel index of the hit, all ones = miss
    uint8_t* shadowed_out;
    unsigned long long* next_item;   // work counter
    unsigned long long* next_zero;   // the NEXT launch's work counter: cleared by this launch (two counters alternate: no memset between traces)
};
struct WalkParams {
    WalkHot hot;
    WalkCold cold;
};
typedef const WalkCold __attribute__((address_space(4)))* WalkColdPtr;
__device__ __forceinline__ WalkColdPtr walk_cold()
{
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (WalkColdPtr)(p + offsetof(WalkParams, cold));
}

#if defined(VX_W_DEBUG) && !defined(VX_W_TS)
#define VX_W_TS
#endif
#ifdef VX_W_TS
__device__ unsigned long long g_walk_ts[3 * 8192];  // per wave: s_memtime at entry, when its queue ran dry, at exit
#endif
#ifdef VX_W_DEBUG
// diagnostic build: lane utilisation per code site.  site i: g_walk_dbg[2i] = times a wave executed the site, [2i+1] = lanes active
// over those executions.  Sites: 0 ray set-up, 1 mip lookup, 2 slab step, 3 brick, 4 candidate slab, 5 exact test, 6 retire, 7 round
// (lanes busy); [16..19] wave cycles in refill / walk / brick / retire.
__device__ unsigned long long g_walk_dbg[24];
__device__ unsigned long long g_walk_hist[2 * 16];  // rays by slab steps (bucket b: [2^b, 2^(b+1)) steps; bucket 0 also holds 0), and the steps they took
#define VX_W_SITE(i) { const unsigned long long m_ = __ballot(true); if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) { dbg_w[i] += 1u; dbg_l[i] += (unsigned)__popcll(m_); } }
#define VX_W_SITE_DECL , unsigned* dbg_w, unsigned* dbg_l
#define VX_W_SITE_ARGS , dbg_w, dbg_l
#define VX_W_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); dbg_c[i] += now_ - dbg_last; dbg_last = now_; }
#else
#define VX_W_SITE(i)
#define VX_W_SITE_DECL
#define VX_W_SITE_ARGS
#define VX_W_T(i)
#endif

#ifndef VOXHIP_DONATE_ON
#define VOXHIP_DONATE_ON (P.donate != 0)
#endif
#ifndef VX_W_DON_BELOW
#define VX_W_DON_BELOW 48
#endif
#ifndef VX_W_DON_BRICKS
#define VX_W_DON_BRICKS 6.0f
#endif
#ifndef VX_W_STEPS
#define VX_W_STEPS 2
#endif
#ifndef VX_W_ITERS
#define VX_W_ITERS 1
#endif
#ifndef VX_W_REFILL
#define VX_W_REFILL 56
#endif
#ifndef VX_W_CHUNK
#define VX_W_CHUNK 64
#endif
#ifndef VX_W_CHUNK_MAX
#define VX_W_CHUNK_MAX 256
#endif
#ifndef VX_W_BLOCK
#define VX_W_BLOCK 256
#endif
#ifndef VX_W_MINWAVES
#define VX_W_MINWAVES 4
#endif

// After a slab has been dealt with: on to the next one -- and up to the block level at the end of a block slab's eight brick
// slabs.  Returns false when the walk leaves the grid.
template <typename IdxT, bool WIDE>
__device__ __forceinline__ bool walk_advance(WalkLane<IdxT, WIDE>& R)
{
    const int s = R.iw > 0.0f ? 1 : -1;
    const int k = R.k;
    int nl = 1, kn = k + s;
    if (R.lvl == 2 || (kn >> 3) != (k >> 3)) { nl = 2; kn = (R.lvl == 2 ? k : (k >> 3)) + s; }
    const int ncw = (R.dimw + (nl == 2 ? 63 : 7)) >> (nl == 2 ? 6 : 3);
    if (kn < 0 || kn >= ncw) return false;  // left the grid
    R.lvl = nl;
    R.k = kn;
    return true;
}

// Level 0, entering a brick: the brick's eight slab words are ONE 64-byte line (four 16-byte loads in flight together -- the
// walk is bound by dependent memory round trips, not by issue); slabs whose word misses the rectangle the ray sweeps across the
// whole brick slab are dropped here without any arithmetic.
template <typename IdxT, bool WIDE>
__device__ __forceinline__ void walk_fetch_brick(WalkLane<IdxT, WIDE>& R, const WalkHot& P VX_W_SITE_DECL)
{
    VX_W_SITE(3)
    const int nu = WIDE ? R.nu_w : (int)((R.pend >> 16) & 0x7FFFu);
    const int j = R.jb + __ffs(R.pend & 0xFFFFu) - 1;
    int jv = 0, ju = j;  // j = jv * nu + ju, by subtraction (nu >= 1)
    while (ju >= nu) { ju -= nu; ++jv; }
    const int cu = pk_lo<WIDE>(R.pc) + ju, cv = pk_hi<WIDE>(R.pc) + jv;
    int bx, by, bz;
    unperm(R.perm, cu, cv, R.k, bx, by, bz);
    const uint32_t bi = (uint32_t)bx + P.d1[0] * ((uint32_t)by + P.d1[1] * (uint32_t)bz);
    const ulonglong2* wp = reinterpret_cast<const ulonglong2*>(P.bricks3 + (uint64_t)R.perm * P.ori_stride + (uint64_t)bi * 8ull);
    const ulonglong2 w01 = wp[0], w23 = wp[1], w45 = wp[2], w67 = wp[3];
    const int bu = cu << 3, bv = cv << 3;
    const int u0 = pk_lo<WIDE>(R.pu), u1 = pk_hi<WIDE>(R.pu), v0 = pk_lo<WIDE>(R.pv), v1 = pk_hi<WIDE>(R.pv);
    const int a0 = (u0 > bu ? u0 : bu) - bu, a1 = (u1 < bu + 7 ? u1 : bu + 7) - bu;  // columns of the brick slab's rectangle inside this brick
    const int b0 = (v0 > bv ? v0 : bv) - bv, b1 = (v1 < bv + 7 ? v1 : bv + 7) - bv;  // rows
    const unsigned long long col = rep8((2u << a1) - (1u << a0));
    const unsigned long long rect = col & (~0ull >> (8 * (7 - b1))) & (~0ull << (8 * b0));
    R.w0 = w01.x; R.w1 = w01.y; R.w2 = w23.x; R.w3 = w23.y; R.w4 = w45.x; R.w5 = w45.y; R.w6 = w67.x; R.w7 = w67.y;
    uint32_t sm = ((w01.x & rect) ? 1u : 0u) | ((w01.y & rect) ? 2u : 0u) | ((w23.x & rect) ? 4u : 0u) | ((w23.y & rect) ? 8u : 0u) |
                  ((w45.x & rect) ? 16u : 0u) | ((w45.y & rect) ? 32u : 0u) | ((w67.x & rect) ? 64u : 0u) | ((w67.y & rect) ? 128u : 0u);
    // slabs wholly in front of the ray's start cell are never looked at
    const int s0 = R.cw0 - (R.k << 3);  // slab of this brick that holds the start cell (negative: in front of the brick)
    if (R.iw > 0.0f) sm &= s0 <= 0 ? 0xFFu : (s0 > 7 ? 0u : (0xFFu << s0));
    else sm &= s0 >= 7 ? 0xFFu : (s0 < 0 ? 0u : (2u << s0) - 1u);
    R.sm = sm & 0xFFu;
    R.pc = pk_make<WIDE>(cu, cv);  // from here on: the brick itself (its rectangle cell is popped from R.pend below)
}

// Level 0, a brick is done (or a brick slab's rectangle has just been found occupied): on to the next occupied brick of the
// rectangle; after the last one back to level 1 and on to the next slab.  Returns false when the ray is finished.
template <typename IdxT, bool WIDE>
__device__ __forceinline__ bool walk_bricks(WalkLane<IdxT, WIDE>& R, const WalkHot& P, bool pop, bool dead VX_W_SITE_DECL)
{
    for (;;) {
        if (pop) {  // R.pc goes back to the rectangle's first brick for the decode of the next one
            const uint32_t low = R.pend & 0xFFFFu;
            const int nu = WIDE ? R.nu_w : (int)((R.pend >> 16) & 0x7FFFu);
            int jv = 0, ju = R.jb + __ffs(low) - 1;  // the finished brick was rectangle cell j = jb + ffs(low) - 1
            while (ju >= nu) { ju -= nu; ++jv; }
            R.pc = pk_make<WIDE>(pk_lo<WIDE>(R.pc) - ju, pk_hi<WIDE>(R.pc) - jv);
            R.pend = (R.pend & 0xFFFF0000u) | (low & (low - 1u));
        }
        if (!(R.pend & 0xFFFFu)) break;
        walk_fetch_brick(R, P VX_W_SITE_ARGS);
        if (R.sm != 0u) return true;
        pop = true;  // nothing of this brick inside the rectangle: on to the next without spending a step
    }
    const bool more = (R.pend >> 31) != 0u;
    R.pend = 0u;
    R.lvl = 1;
    if (more) { R.jb += 16; return true; }  // the same brick slab's next window of rectangle cells
    if (dead) return false;
    R.jb = 0;
    return walk_advance(R);
}

// One step of the walk = one slab: a block slab (level 2), a brick slab (level 1), or one 1-cell slab of the brick being walked
// (level 0).  All three share the slab's [ta, tb], the termination test and the rectangle of cells; they differ in what the
// rectangle is looked up in.  Returns false when the ray is finished.
template <bool LDS_MIPS, typename IdxT, bool WIDE>
__device__ __forceinline__ bool walk_step(WalkLane<IdxT, WIDE>& R, const WalkHot& P, const uint32_t* __restrict__ mips_lds, bool& xpend, uint4* __restrict__ xslot,
                                          uint32_t* __restrict__ xslot_hi /*WIDE: the upper half of the parked brick coordinates*/, uint32_t& fstate VX_W_SITE_DECL)
{
    VX_W_SITE(2)
    const GridParams& g = P.g;
    const int lvl = R.lvl;
    const bool pos = R.iw > 0.0f;
    // ---- which slab
    int sl = 0;
    const bool brick_empty = lvl == 0 && R.sm == 0u;  // nothing (left) of this brick inside the rectangle
    if (lvl == 0 && !brick_empty) {
        sl = pos ? (__ffs(R.sm) - 1) : (31 - __clz(R.sm));  // next candidate slab of the brick in travel order
        R.sm &= ~(1u << sl);
    }
    const int sh = lvl == 2 ? 6 : (lvl == 1 ? 3 : 0);
    const int k = lvl == 0 ? (R.k << 3) + sl : R.k;
    const int i0 = k << sh;
    int i1 = (k + 1) << sh;
    i1 = i1 > R.dimw ? R.dimw : i1;
    // ---- the slab's [ta, tb] along the major axis
    const float lo = (R.orgw + (float)i0 * g.vs) - R.tol, hi = (R.orgw + (float)i1 * g.vs) + R.tol;
    const float t_lo = R.iw * (lo - R.ow), t_hi = R.iw * (hi - R.ow);
    const float ta = pos ? t_lo : t_hi, tb = pos ? t_hi : t_lo;
    const bool stop = brick_empty || R.best < ta || ta > R.tf;  // no box of this slab or any later one can beat the best hit / beyond the interval
    if (stop && lvl != 0) return false;
    // (level 0: only this brick is done -- another brick of the same brick slab may still hold an earlier cell)
    const float ca = fmaxf(ta, R.tn), cb = fminf(tb, R.tf);
    bool brick_done = lvl == 0 && (stop || R.sm == 0u);
    if (ca <= cb && !stop) {
        // ---- the rectangle of cells the ray can touch inside the slab
        const float tol2 = 2.0f * R.tol;
        const float ua = R.ou + ca * R.du, ub = R.ou + cb * R.du;
        const float va = R.ov + ca * R.dv, vb = R.ov + cb * R.dv;
        int u0 = (int)floorf(((fminf(ua, ub) - tol2) - R.orgu) * P.inv_vs), u1 = (int)floorf(((fmaxf(ua, ub) + tol2) - R.orgu) * P.inv_vs);
        int v0 = (int)floorf(((fminf(va, vb) - tol2) - R.orgv) * P.inv_vs), v1 = (int)floorf(((fmaxf(va, vb) + tol2) - R.orgv) * P.inv_vs);
        u0 = u0 < 0 ? 0 : u0;
        v0 = v0 < 0 ? 0 : v0;
        u1 = u1 > R.dimu - 1 ? R.dimu - 1 : u1;
        v1 = v1 > R.dimv - 1 ? R.dimv - 1 : v1;
        if (lvl == 0) {
            // ---- level 0: the rectangle inside the brick as a bit mask on the slab's word; exact tests on what survives
            // (selection BY VALUE: `c ? R.a : R.b` on struct members is an lvalue conditional -- clang selects the address and the lane
            // state ends up in scratch memory)
            const unsigned long long bits = sel8(sl, R.w0, R.w1, R.w2, R.w3, R.w4, R.w5, R.w6, R.w7);
            const int bu = pk_lo<WIDE>(R.pc) << 3, bv = pk_hi<WIDE>(R.pc) << 3;
            int a0 = u0 - bu, a1 = u1 - bu, b0 = v0 - bv, b1 = v1 - bv;
            a0 = a0 < 0 ? 0 : a0; b0 = b0 < 0 ? 0 : b0;
            a1 = a1 > 7 ? 7 : a1; b1 = b1 > 7 ? 7 : b1;
            unsigned long long cand = 0ull;
            if (a0 <= a1 && b0 <= b1) {
                const unsigned long long col = rep8((2u << a1) - (1u << a0));
                cand = bits & col & (~0ull >> (8 * (7 - b1))) & (~0ull << (8 * b0));
            }
            if (cand) {
                // The exact tests are not done here, where one lane in sixteen has candidates: the lane parks them in its LDS slot
                // and sits out the rest of the round; all parked lanes of the wave test together at the end of the round (walk_exact).
                // Until then the lane's best hit is stale, which only postpones the pruning that depends on it.
                *xslot = make_uint4((uint32_t)cand, (uint32_t)(cand >> 32), (uint32_t)k, (uint32_t)R.pc);
                if (WIDE) *xslot_hi = (uint32_t)((unsigned long long)R.pc >> 32);
                xpend = true;
            }
        } else {
            const bool top = lvl == 2;
            const int cu0 = u0 >> sh, cv0 = v0 >> sh;
            const int nu = (u1 >> sh) - cu0 + 1, nv = (v1 >> sh) - cv0 + 1;  // (u1, v1 may be -1: the arithmetic shift keeps them negative)
            const int n = (nu > 0 && nv > 0) ? nu * nv : 0;                  // 1 .. 4 cells of this level as a rule, 9 at most
            const uint32_t Dx = top ? P.d2[0] : P.d1[0], Dy = top ? P.d2[1] : P.d1[1];
            const uint32_t moff = top ? P.m1_words : 0u;
            // The rectangle is looked at in windows of 16 cells.  One window is all there ever is unless the position tolerance is
            // comparable to a brick (coordinates of ~10^6 voxel sizes, where float32 no longer resolves the voxels anyway); the
            // window logic only keeps that regime correct, not fast.
            const int jb = R.jb;
            uint32_t occ = 0u;  // bit j - jb: cell (cu0 + j % nu, cv0 + j / nu) of the rectangle is occupied
            int jend;
            if (jb == 0 && nu <= 2 && nv <= 2) {
                // The rule -- a rectangle of at most 2 x 2 cells: its four look-ups side by side instead of a loop that every lane
                // of the wave sits through for as many turns as the largest rectangle among them needs.  A cell the rectangle does
                // not have reads the first cell's word again and is masked out.
                VX_W_SITE(1)
                jend = n;
                int x, y, z;
                unperm(R.perm, cu0, cv0, k, x, y, z);
                const uint32_t i00 = n > 0 ? (uint32_t)x + Dx * ((uint32_t)y + Dy * (uint32_t)z) : 0u;  // (an empty rectangle may lie outside the grid)
                // linear-index strides of one cell along u and along v:  w=0: u=y v=z;  w=1: u=z v=x;  w=2: u=x v=y
                const uint32_t su = R.perm == 0 ? Dx : (R.perm == 1 ? Dx * Dy : 1u), sv = R.perm == 0 ? Dx * Dy : (R.perm == 1 ? 1u : Dx);
                const bool hu = nu == 2, hv = nv == 2;
                const uint32_t i10 = hu ? i00 + su : i00, i01 = hv ? i00 + sv : i00, i11 = (hu && hv) ? i00 + su + sv : i00;
                uint32_t w00, w10, w01, w11;
                if (LDS_MIPS) {
                    w00 = mips_lds[moff + (i00 >> 5)]; w10 = mips_lds[moff + (i10 >> 5)]; w01 = mips_lds[moff + (i01 >> 5)]; w11 = mips_lds[moff + (i11 >> 5)];
                } else {
                    const uint32_t* __restrict__ mp = top ? P.w2 : P.w1;
                    w00 = mp[i00 >> 5]; w10 = mp[i10 >> 5]; w01 = mp[i01 >> 5]; w11 = mp[i11 >> 5];
                }
                const uint32_t b00 = (w00 >> (i00 & 31u)) & 1u, b10 = (w10 >> (i10 & 31u)) & 1u, b01 = (w01 >> (i01 & 31u)) & 1u, b11 = (w11 >> (i11 & 31u)) & 1u;
                if (n > 0) occ = b00 | (hu ? b10 << 1 : 0u) | (hv ? b01 << nu : 0u) | ((hu && hv) ? b11 << 3 : 0u);
            } else {
                int ju = jb, jv = 0;
                while (ju >= nu && nu > 0) { ju -= nu; ++jv; }
                jend = n < jb + 16 ? n : jb + 16;
#pragma nounroll
                for (int j = jb; j < jend; ++j) {
                    VX_W_SITE(1)
                    int x, y, z;
                    unperm(R.perm, cu0 + ju, cv0 + jv, k, x, y, z);
                    const uint32_t i = (uint32_t)x + Dx * ((uint32_t)y + Dy * (uint32_t)z);
                    const uint32_t wd = LDS_MIPS ? mips_lds[moff + (i >> 5)] : (top ? P.w2[i >> 5] : P.w1[i >> 5]);
                    occ |= ((wd >> (i & 31u)) & 1u) << (j - jb);
                    if (++ju == nu) { ju = 0; ++jv; }
                }
            }
            const bool more = jend < n;
            if (occ) {
                if (top) {  // down into the eight brick slabs of this block slab
                    const int ncw = (R.dimw + 7) >> 3;
                    // the first brick slab of this block slab -- not in front of the ray's start cell
                    const int ks = R.cw0 >> 3;
                    int kf = pos ? ((k << 3) > ks ? (k << 3) : ks) : ((k << 3) + 7 < ks ? (k << 3) + 7 : ks);
                    kf = kf < (k << 3) ? (k << 3) : (kf > (k << 3) + 7 ? (k << 3) + 7 : kf);
                    kf = kf > ncw - 1 ? ncw - 1 : kf;
                    R.lvl = 1;
                    R.k = kf;
                    R.jb = 0;
                    return true;
                }
                // the occupied bricks of this brick slab's rectangle: walked one after the other at level 0
                R.pend = occ | (WIDE ? 0u : ((uint32_t)nu << 16)) | (more ? 0x80000000u : 0u);
                if (WIDE) R.nu_w = nu;
                R.pc = pk_make<WIDE>(cu0, cv0);
                R.pu = pk_make<WIDE>(u0, u1);
                R.pv = pk_make<WIDE>(v0, v1);
                R.lvl = 0;
                R.sm = 0u;
                brick_done = true;  // "no brick loaded yet": fetch the first one below
            } else if (more) {  // next window of the same slab
                R.jb = jend;
                return true;
            }
        }
    }
    if (R.lvl == 0) {
        // ---- level 0 bookkeeping: next brick of the brick slab's rectangle once this one is done; back to level 1 after the last
        if (!brick_done) return true;
        // `dead`: this step found that no box of its slab or of any later slab can beat the best hit; only the sibling bricks of the
        // same brick slab lie beside and not behind.  Without one the ray is finished -- no step is spent on rediscovering that at
        // the next brick slab and the next block slab.
        const bool dead = stop && !brick_empty;
        const bool pop = lvl == 0;  // a brick was being walked
        // The brick fetch is not done here, where one lane in eight needs it: the lane sits out the rest of the round and all such
        // lanes of the wave fetch together at its end (walk_bricks; measured -4 % at 1M and at 8M rays).
        fstate = 1u | (pop ? 2u : 0u) | (dead ? 4u : 0u);
        return true;
    }
    R.jb = 0;
    return walk_advance(R);
}

// The exact tests of the candidates a lane parked in its LDS slot: hitAabb on every candidate cell of one 1-cell slab of a brick.
template <typename IdxT, bool WIDE>
__device__ __forceinline__ void walk_exact(WalkLane<IdxT, WIDE>& R, const WalkHot& P, const uint4* __restrict__ xslot, const uint32_t* __restrict__ xslot_hi VX_W_SITE_DECL)
{
    const GridParams& g = P.g;
    const uint4 xs = *xslot;
    unsigned long long cand = (unsigned long long)xs.x | ((unsigned long long)xs.y << 32);
    const int k = (int)xs.z;
    const int bu = (WIDE ? (int)xs.w : (int)(xs.w & 0xFFFFu)) << 3, bv = (WIDE ? (int)*xslot_hi : (int)(xs.w >> 16)) << 3;
    // the slab's own box planes along w (voxelgridBool.cpp:37-41: c -/+ half with c = org + (i + 0.5) * vs) and their hitAabb terms
    const float cw = R.orgw + (((float)k + 0.5f) * g.vs);
    const float bw = R.iw * ((cw - g.half) - R.ow), tw = R.iw * ((cw + g.half) - R.ow);
    const float mnw = fminf(tw, bw), mxw = fmaxf(tw, bw);
    while (cand) {
        VX_W_SITE(5)
        const int b = __ffsll((long long)cand) - 1;
        cand &= cand - 1ull;
        const int cu_ = bu + (b & 7), cv_ = bv + (b >> 3);
        const float ccu = R.orgu + (((float)cu_ + 0.5f) * g.vs), ccv = R.orgv + (((float)cv_ + 0.5f) * g.vs);
        const float bu_ = R.iu * ((ccu - g.half) - R.ou), tu_ = R.iu * ((ccu + g.half) - R.ou);
        const float bv_ = R.iv * ((ccv - g.half) - R.ov), tv_ = R.iv * ((ccv + g.half) - R.ov);
        // hitAabb, rint:46-56: t0 = max of the per-axis minima, t1 = min of the per-axis maxima (any axis order: same floats)
        const float t0 = fmaxf(fmaxf(fminf(tu_, bu_), fminf(tv_, bv_)), mnw);
        const float t1 = fminf(fminf(fmaxf(tu_, bu_), fmaxf(tv_, bv_)), mxw);
        const float t = t1 > fmaxf(t0, 0.0f) ? t0 : -1.0f;
        int cx, cy, cz;
        unperm(R.perm, cu_, cv_, k, cx, cy, cz);
        const IdxT vi = (IdxT)(uint32_t)cx + (IdxT)g.dim[0] * ((IdxT)(uint32_t)cy + (IdxT)g.dim[1] * (IdxT)(uint32_t)cz);
        if (t > 0.0f && t >= P.tmin && t <= R.tmax &&                 // rint:69, rgen:50-51
            (t < R.best || (t == R.best && vi < R.best_idx))) {
            R.best = t;
            R.best_idx = vi;
        }
    }
}

// position of the n-th (0-based) set bit of a 64-bit mask (n < popcount)
__device__ __forceinline__ int nth_set_bit64_w(unsigned long long m, int n)
{
    int pos = 0;
    unsigned lo = (unsigned)m;
    int c = __popc(lo);
    if (n >= c) { n -= c; pos = 32; lo = (unsigned)(m >> 32); }
    c = __popc(lo & 0xFFFFu); if (n >= c) { n -= c; pos += 16; lo >>= 16; }
    c = __popc(lo & 0xFFu);   if (n >= c) { n -= c; pos += 8;  lo >>= 8; }
    c = __popc(lo & 0xFu);    if (n >= c) { n -= c; pos += 4;  lo >>= 4; }
    c = __popc(lo & 0x3u);    if (n >= c) { n -= c; pos += 2;  lo >>= 2; }
    c = lo & 1u;              if (n >= c) { pos += 1; }
    return pos;
}
__device__ __forceinline__ float shfl_fw(float v, int src) { return __shfl(v, src, 64); }

// Persistent waves with dynamic work fetch.  Exit condition every wave reaches: the work counter passes the ray count (no
// refill possible) and every lane's ray has finished; a ray finishes in a bounded number of slabs (its index along the major
// axis is monotone at both levels).
template <bool LDS_MIPS, typename IdxT, bool WIDE = false>
__global__ __launch_bounds__(VX_W_BLOCK, VX_W_MINWAVES) void k_walk(const WalkParams PP)
{
    const WalkHot& P = PP.hot;
    constexpr int kStepsPerRound = VX_W_STEPS;   // slab steps between two brick phases
    constexpr int kItersPerRound = VX_W_ITERS;   // (walk, brick) iterations between two refill checks
    constexpr int kRefillBelow = VX_W_REFILL;    // refill when fewer than this many lanes are busy
    constexpr uint32_t kChunkRays = VX_W_CHUNK;      // rays a wave reserves per touch of the global counter: at least ...
    constexpr uint32_t kChunkMax = VX_W_CHUNK_MAX;   // ... and at most
    constexpr uint32_t kWavesPerBlock = VX_W_BLOCK / 64;
    const GridParams& g = P.g;
    extern __shared__ __attribute__((aligned(16))) uint32_t mips_lds[];
    if (blockIdx.x == 0 && threadIdx.x == 0) *walk_cold()->next_zero = 0ull;
    if (LDS_MIPS) {
        for (uint32_t i = threadIdx.x; i < P.m1_words; i += VX_W_BLOCK) mips_lds[i] = P.w1[i];
        for (uint32_t i = threadIdx.x; i < P.m2_words; i += VX_W_BLOCK) mips_lds[P.m1_words + i] = P.w2[i];
        __syncthreads();
    }
    // Work donation (drain phase): the pieces of a split ray stay inside their wave and meet in LDS -- a 64-bit minimum of
    // (t bits << 32 | voxel index), the same "closest, then lower index" order a single walk uses, and a piece count; the piece
    // that finishes last writes the ray's outputs.  Slot = the lane that held the ray when it was first split.
    constexpr bool kDonate = sizeof(IdxT) == 4;       // the merge key holds the voxel index in 32 bits
    constexpr int kDonateBelow = VX_W_DON_BELOW;       // donate while at most this many lanes are busy
    constexpr float kDonateBricks = VX_W_DON_BRICKS;  // ... and only from pieces with more than this many brick slabs left
    __shared__ unsigned long long don_key[VX_W_BLOCK];
    __shared__ unsigned don_cnt[VX_W_BLOCK];
    __shared__ uint4 xslots[VX_W_BLOCK];  // per lane: the candidate cells parked for the round's exact-test phase
    __shared__ uint32_t xslots_hi[WIDE ? VX_W_BLOCK : 1];
    bool xpend = false;
    uint32_t fstate = 0u;  // bit 0: the lane waits for the round's brick phase; bits 1, 2: walk_bricks' pop and dead
    const int lane = threadIdx.x & 63;
    int slot = -1;             // >= 0: this lane walks a PIECE of a split ray; its result goes through don_key[slot]
    WalkLane<IdxT, WIDE> R;
    R.pend = 0u;
#ifdef VX_W_DEBUG
    unsigned dbg_w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dbg_c[4] = {0, 0, 0, 0}, dbg_last = __builtin_readcyclecounter();
    unsigned dbg_steps = 0;
#endif
#ifdef VX_W_TS
    const unsigned long long ts_begin = __builtin_readcyclecounter();
    unsigned long long ts_drained = 0;
#endif
    uint32_t r = 0xFFFFFFFFu;  // ray this lane is tracing
    bool busy = false;         // traversal in progress
    bool drained = false;      // no ray left for this wave: the global counter and the wave's chunk are exhausted
    bool drained_global = false;
    uint32_t chunk_cur = 0, chunk_end = 0;
    const uint32_t nrays = P.nrays;
    // The first chunk of every wave is assigned statically: thousands of waves asking the one counter in the same microsecond
    // queue up behind each other at the memory-side atomic unit.
    uint32_t static_rays;
    {
        uint32_t csz = nrays / (2u * kWavesPerBlock * gridDim.x);
        csz = csz > kChunkMax ? kChunkMax : csz;
        csz = csz < kChunkRays ? kChunkRays : (csz & ~63u);
        const uint64_t sr = (uint64_t)csz * kWavesPerBlock * gridDim.x, c0 = (uint64_t)csz * ((uint64_t)kWavesPerBlock * blockIdx.x + (threadIdx.x >> 6));
        static_rays = sr > nrays ? nrays : (uint32_t)sr;
        chunk_cur = c0 > nrays ? nrays : (uint32_t)c0;
        chunk_end = c0 + csz > nrays ? nrays : (uint32_t)(c0 + csz);
        if (sr >= nrays) drained_global = true;
    }
    for (;;) {
        const unsigned long long busy_mask = __ballot(busy);
        const int nbusy = __popcll(busy_mask);
        if (!drained && nbusy < kRefillBelow) {
            // ---- refill idle lanes.  Ray indices come from a per-wave chunk; the global counter is touched once per chunk.
            const unsigned long long idle_mask = ~busy_mask;
            const uint32_t need = (uint32_t)(64 - nbusy);
            const uint32_t take = need < chunk_end - chunk_cur ? need : chunk_end - chunk_cur;
            const uint32_t first = chunk_cur;
            chunk_cur += take;
            uint32_t second = 0xFFFFFFFFu;
            if (take < need && !drained_global) {
                // Guided chunk size: the wave waits ~2 us for the counter's old value, so it asks for a large chunk while much is
                // left (half an even share of what remained at its previous fetch) and for the minimum near the end.
                const uint32_t left = nrays > chunk_end ? nrays - chunk_end : 0u;
                uint32_t csz = left / (2u * kWavesPerBlock * gridDim.x);
                csz = csz > kChunkMax ? kChunkMax : csz;
                csz = csz < kChunkRays ? kChunkRays : (csz & ~63u);
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(walk_cold()->next_item, (unsigned long long)csz);
                base = (((unsigned long long)__shfl((unsigned)(base >> 32), 0, 64) << 32) | __shfl((unsigned)base, 0, 64)) + static_rays;
                if (base < nrays) {
                    second = (uint32_t)base;
                    chunk_cur = second + (need - take);
                    chunk_end = base + csz > nrays ? nrays : (uint32_t)(base + csz);
                    if (chunk_cur > chunk_end) chunk_cur = chunk_end;
                }
                if (base + csz >= nrays) drained_global = true;
            }
            if (drained_global && chunk_cur >= chunk_end) drained = true;
#ifdef VX_W_TS
            if (drained && !ts_drained) ts_drained = __builtin_readcyclecounter();
#endif
            if (!busy) {
                const uint32_t posn = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                uint32_t mine = 0xFFFFFFFFu;
                if (posn < take) mine = first + posn;
                else if (second != 0xFFFFFFFFu && second + (posn - take) < chunk_end) mine = second + (posn - take);
                if (mine != 0xFFFFFFFFu) {
                    VX_W_SITE(0)
                    r = mine;
                    const WalkColdPtr C = walk_cold();
                    const uint64_t ro = C->ray_base + r;
                    const float* rays = C->rays;
                    float ox, oy, oz, dx, dy, dz;
                    load_ray_w(rays == nullptr, ro, rays, C->cam, ox, oy, oz, dx, dy, dz);
                    const float* tpr = C->tmax_per_ray;
                    const float tmax_r = tpr ? tpr[ro] : C->tmax;
                    busy = walk_setup(R, g, P.inv_vs, tmax_r, ox, oy, oz, dx, dy, dz);
                    if (!busy) {  // cannot touch the grid: retire at once
                        float* t_out = C->t_out;
                        IdxT* idx_out = (IdxT*)C->idx_out;
                        uint8_t* shadowed_out = C->shadowed_out;
                        if (t_out) t_out[ro] = -1.0f;
                        if (idx_out) idx_out[ro] = (IdxT)~(IdxT)0;
                        if (shadowed_out) shadowed_out[ro] = 0;
                    }
                }
            }
        }
        VX_W_T(0)
        if (!__ballot(busy)) {
            if (drained) break;
            continue;
        }
        if (busy) { VX_W_SITE(7) }
        // ---- work donation (drain phase).  With a few rays per lane the queue empties early and every wave is left with a few
        // long rays on a shrinking set of lanes.  Once no new ray can be fetched, a busy lane with enough of its t interval left
        // hands the FAR half to an idle lane of its wave (the ray's state travels by shuffles); each piece walks the slabs that
        // overlap its part of the interval (the slab that holds the cut is looked at by both).
        if (kDonate && drained) {
            const unsigned long long bm = __ballot(busy);
            const int nb = __popcll(bm);
            if (nb <= kDonateBelow && VOXHIP_DONATE_ON) {
                const bool pos = R.iw > 0.0f;
                const int sh = R.lvl == 2 ? 6 : 3;
                int ip = pos ? (R.k << sh) : ((R.k + 1) << sh);
                ip = ip > R.dimw ? R.dimw : ip;
                const float t_cur = fmaxf(R.iw * (((R.orgw + (float)ip * g.vs) + (pos ? -R.tol : R.tol)) - R.ow), R.tn);  // the current slab's ta
                const float t_end = fminf(R.tf, R.best);
                const bool can = busy && R.lvl != 0 && (t_end - t_cur > kDonateBricks * 8.0f * g.vs * fabsf(R.iw));
                const unsigned long long dm = __ballot(can);
                const unsigned long long im = ~bm;
                const int ndon = min(__popcll(dm), 64 - nb);
                if (ndon > 0) {
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    const bool donor = can && __popcll(dm & lt) < ndon;
                    const float t_mid = 0.5f * (t_cur + t_end);
                    if (donor) {
                        if (slot < 0) {  // first split of this ray: open its merge slot (this piece counts as one)
                            slot = (int)threadIdx.x;
                            don_key[slot] = ~0ull;
                            don_cnt[slot] = 1u;
                        }
                        atomicAdd(&don_cnt[slot], 1u);  // the piece handed out below
                    }
                    const int irank = __popcll(im & lt);
                    const bool recv = !busy && irank < ndon;
                    const int src = nth_set_bit64_w(dm, recv ? irank : 0);
                    const float s_ou = shfl_fw(R.ou, src), s_ov = shfl_fw(R.ov, src), s_ow = shfl_fw(R.ow, src), s_du = shfl_fw(R.du, src), s_dv = shfl_fw(R.dv, src);
                    const float s_iu = shfl_fw(R.iu, src), s_iv = shfl_fw(R.iv, src), s_iw = shfl_fw(R.iw, src);
                    const float s_orgu = shfl_fw(R.orgu, src), s_orgv = shfl_fw(R.orgv, src), s_orgw = shfl_fw(R.orgw, src);
                    const int s_dimu = __shfl(R.dimu, src, 64), s_dimv = __shfl(R.dimv, src, 64), s_dimw = __shfl(R.dimw, src, 64), s_perm = __shfl(R.perm, src, 64);
                    const float s_tol = shfl_fw(R.tol, src), s_tmax = shfl_fw(R.tmax, src), s_best = shfl_fw(R.best, src), s_mid = shfl_fw(t_mid, src), s_end = shfl_fw(t_end, src);
                    const uint32_t s_bidx = __shfl((uint32_t)R.best_idx, src, 64), s_r = __shfl(r, src, 64);
                    const int s_slot = __shfl(slot, src, 64);
                    if (donor) R.tf = t_mid;  // keep the near half
                    if (recv) {
                        R.ou = s_ou; R.ov = s_ov; R.ow = s_ow; R.du = s_du; R.dv = s_dv; R.iu = s_iu; R.iv = s_iv; R.iw = s_iw;
                        R.orgu = s_orgu; R.orgv = s_orgv; R.orgw = s_orgw; R.dimu = s_dimu; R.dimv = s_dimv; R.dimw = s_dimw; R.perm = s_perm;
                        R.tol = s_tol; R.tmax = s_tmax; R.best = s_best; R.best_idx = (IdxT)s_bidx;
                        R.tn = s_mid; R.tf = s_end;
                        R.pend = 0u; R.jb = 0; R.sm = 0u; R.lvl = 2;
                        // first block slab of the piece: the one that holds the cut (one cell of slack, as in walk_setup)
                        const float pw = s_ow + s_mid * (1.0f / s_iw);
                        int cw = (int)floorf((pw - s_orgw) * P.inv_vs) + (s_iw > 0.0f ? -1 : 1);
                        cw = cw < 0 ? 0 : (cw > s_dimw - 1 ? s_dimw - 1 : cw);
                        R.k = cw >> 6;
                        R.cw0 = cw;
                        r = s_r;
                        slot = s_slot;
                        busy = true;
                    }
                }
            }
        }
        // ---- walk: every busy lane advances by one slab per step, whatever its level
        bool finished = false;
        for (int s = 0; s < kStepsPerRound * kItersPerRound; ++s) {
            const bool go = busy && !finished && !xpend && !fstate;
            if (!__ballot(go)) break;
#ifdef VX_W_DEBUG
            if (go) ++dbg_steps;
#endif
            if (go && !walk_step<LDS_MIPS>(R, P, mips_lds, xpend, &xslots[threadIdx.x], &xslots_hi[WIDE ? threadIdx.x : 0], fstate VX_W_SITE_ARGS)) finished = true;
        }
        // ---- exact tests of the candidates the steps parked
        if (__ballot(xpend)) {
            if (xpend) {
                walk_exact(R, P, &xslots[threadIdx.x], &xslots_hi[WIDE ? threadIdx.x : 0] VX_W_SITE_ARGS);
                xpend = false;
                if (P.any_hit && R.best_idx != (IdxT)~(IdxT)0) finished = true;  // shadow query (gl_RayFlagsTerminateOnFirstHitEXT, raytrace2.rchit:108)
            }
        }
        // ---- brick phase: the lanes whose step ended at a brick boundary fetch their next brick
        if (__ballot(fstate != 0u)) {
            if (fstate) {
                if (!finished && !walk_bricks(R, P, (fstate & 2u) != 0u, (fstate & 4u) != 0u VX_W_SITE_ARGS)) finished = true;
                fstate = 0u;
            }
        }
        VX_W_T(1)
        // ---- retire: t and the voxel index of the hit; the primitive rank (two dependent loads), the normal and the hit
        // compaction are done by k_rank over all rays afterwards, off this kernel's critical path
        if (__ballot(finished)) {
            const WalkColdPtr C = walk_cold();
            if (finished) {
                VX_W_SITE(6)
#ifdef VX_W_DEBUG
                {
                    const int bk = dbg_steps ? 31 - __clz(dbg_steps) : 0;
                    atomicAdd(&g_walk_hist[bk < 15 ? bk : 15], 1ull);
                    atomicAdd(&g_walk_hist[16 + (bk < 15 ? bk : 15)], (unsigned long long)dbg_steps);
                    dbg_steps = 0;
                }
#endif
                bool hit = R.best_idx != (IdxT)~(IdxT)0;
                float best_t = R.best;
                IdxT best_i = R.best_idx;
                bool write = true;
                if (kDonate && slot >= 0) {
                    // a piece of a split ray: merge, and only the piece that finishes last reports
                    if (hit) atomicMin(&don_key[slot], ((unsigned long long)__float_as_uint(R.best) << 32) | (unsigned long long)(uint32_t)R.best_idx);
                    write = atomicSub(&don_cnt[slot], 1u) == 1u;
                    if (write) {
                        const unsigned long long key = don_key[slot];
                        hit = key != ~0ull;
                        best_t = __uint_as_float((uint32_t)(key >> 32));
                        best_i = hit ? (IdxT)(uint32_t)key : (IdxT)~(IdxT)0;
                    }
                    slot = -1;
                }
                if (write) {
                    const uint64_t ro = C->ray_base + r;
                    float* t_out = C->t_out;
                    IdxT* idx_out = (IdxT*)C->idx_out;
                    uint8_t* shadowed_out = C->shadowed_out;
                    if (t_out) t_out[ro] = hit ? best_t : -1.0f;
                    if (idx_out) idx_out[ro] = best_i;
                    if (shadowed_out) shadowed_out[ro] = hit ? 1 : 0;
                }
                busy = false;
            }
        }
        VX_W_T(3)
    }
#ifdef VX_W_DEBUG
    for (int i = 0; i < 8; ++i) {
        unsigned a = dbg_w[i], b = dbg_l[i];
        for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
        if (lane == 0) { atomicAdd(&g_walk_dbg[2 * i], (unsigned long long)a); atomicAdd(&g_walk_dbg[2 * i + 1], (unsigned long long)b); }
    }
    if (lane == 0) for (int i = 0; i < 4; ++i) atomicAdd(&g_walk_dbg[16 + i], dbg_c[i]);
#endif
#ifdef VX_W_TS
    {
        const uint32_t wid = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
        if (lane == 0 && wid < 8192u) { g_walk_ts[3 * wid] = ts_begin; g_walk_ts[3 * wid + 1] = ts_drained; g_walk_ts[3 * wid + 2] = __builtin_readcyclecounter(); }
    }
#endif
}

#ifdef VX_W_DEBUG
}  // namespace vx
extern "C" int vx_debug_walk(unsigned long long* out24, int reset)
{
    if (out24 && hipMemcpyFromSymbol(out24, HIP_SYMBOL(vx::g_walk_dbg), 24 * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[24] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(vx::g_walk_dbg), z, 24 * 8) != hipSuccess) return 1; }
    return 0;
}
extern "C" int vx_debug_walk_hist(unsigned long long* out32, int reset)
{
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(vx::g_walk_hist), 32 * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(vx::g_walk_hist), z, 32 * 8) != hipSuccess) return 1; }
    return 0;
}

namespace vx {
#endif
#ifdef VX_W_TS
}  // namespace vx
extern "C" int vx_debug_walk_ts(unsigned long long* out /*[3 * 8192]*/)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vx::g_walk_ts), 3 * 8192 * 8) == hipSuccess ? 0 : 1;
}
namespace vx {
#endif

void launch_walk(const GridParams& g, const TraceMips& mips, const TraceIO& io, unsigned long long* counters /*[2], both zero before the first launch*/,
                 int* phase /*host: which of the two the next launch draws from*/, void* idx_out, bool idx32, hipStream_t s, WalkQueue* queue)
{
    if (!io.nrays) return;
    const uint64_t n1 = (uint64_t)mips.d1[0] * mips.d1[1] * mips.d1[2], n2 = (uint64_t)mips.d2[0] * mips.d2[1] * mips.d2[2];
    const uint32_t m1_words = (uint32_t)((n1 + 31) / 32), m2_words = (uint32_t)((n2 + 31) / 32);
    // VOXHIP_TRACE_LDS=0 forces the global-memory mips (the path every grid above ~550^3 takes) -- used by the parity tests
    const char* env_lds = getenv("VOXHIP_TRACE_LDS");
    const bool wide = g.dim[0] > 65535u || g.dim[1] > 65535u || g.dim[2] > 65535u;  // cell coordinates beyond 16 bits: the WIDE variants (global-memory mips only)
    const bool lds = !wide && (size_t)(m1_words + m2_words) * 4 <= 32768 + 1024 && !(env_lds && atoi(env_lds) == 0);
    const int env_blocks = getenv("VOXHIP_TRACE_BLOCKS") ? atoi(getenv("VOXHIP_TRACE_BLOCKS")) : 0;
    WalkParams P;
    std::memset(&P, 0, sizeof(P));
    P.hot.g = g;
    P.hot.bricks3 = mips.bricks3;
    P.hot.ori_stride = n1 * 8ull;
    P.hot.w1 = mips.w1;
    P.hot.w2 = mips.w2;
    for (int a = 0; a < 3; ++a) { P.hot.d1[a] = mips.d1[a]; P.hot.d2[a] = mips.d2[a]; }
    P.hot.inv_vs = 1.0f / g.vs;
    P.hot.tmin = io.tmin;
    P.hot.any_hit = io.any_hit ? 1 : 0;
    P.hot.m1_words = m1_words;
    P.hot.m2_words = m2_words;
    P.hot.donate = (getenv("VOXHIP_TRACE_DONATE") ? atoi(getenv("VOXHIP_TRACE_DONATE")) : 1) ? 1 : 0;
    P.cold.rays = io.rays;
    P.cold.cam = io.cam_dev;
    P.cold.tmax_per_ray = io.tmax_per_ray;
    P.cold.tmax = io.tmax;
    P.cold.t_out = io.t_out;
    P.cold.idx_out = idx_out;
    P.cold.shadowed_out = io.shadowed_out;

    const size_t shmem = lds ? (size_t)(m1_words + m2_words) * 4 : 0;
    // one launch per 2^31 rays: ray indices inside the kernel are 32-bit
    for (uint64_t base = 0; base < io.nrays; base += 0x80000000ull) {
        const uint64_t n = io.nrays - base < 0x80000000ull ? io.nrays - base : 0x80000000ull;
        P.hot.nrays = (uint32_t)n;
        P.cold.ray_base = base;
        P.cold.next_item = counters + (*phase & 1);
        P.cold.next_zero = counters + ((*phase & 1) ^ 1);
        *phase ^= 1;
        uint64_t max_blocks = 256ull * VX_W_MINWAVES * 4ull * 64ull / VX_W_BLOCK;  // one resident set of waves
        // Batches of a few rays per lane: the launch ends with every wave draining the rays it started last, a span of about two mean
        // ray latencies whatever the batch; with somewhat fewer waves a lane sees at least ~4 rays and the same work drains from fewer
        // rays in flight (measured on the bench scene, 0.5M / 0.75M / 1M rays: -3 / -8 / -8 %; below ~0.4M rays, where a lane has at
        // most one or two rays anyway, the full width is faster, and from ~1.1M rays on the rule gives the full width).
        if (n >= 400000ull && n / 1040ull < max_blocks) max_blocks = n / 1040ull;
        if (env_blocks > 0) max_blocks = (uint64_t)env_blocks;
        uint64_t nblk = (n + VX_W_BLOCK - 1) / VX_W_BLOCK;
        if (nblk > max_blocks) nblk = max_blocks;
        const dim3 grid((unsigned)nblk), block(VX_W_BLOCK);
        if (queue) {  // the work counter of this launch and the value it reaches when the last chunk of rays has been handed out (the kernel's own arithmetic)
            uint32_t csz = (uint32_t)n / (2u * (VX_W_BLOCK / 64u) * (uint32_t)nblk);
            csz = csz > (uint32_t)VX_W_CHUNK_MAX ? (uint32_t)VX_W_CHUNK_MAX : csz;
            csz = csz < (uint32_t)VX_W_CHUNK ? (uint32_t)VX_W_CHUNK : (csz & ~63u);
            const uint64_t sr = (uint64_t)csz * (VX_W_BLOCK / 64u) * nblk;
            queue->counter = P.cold.next_item;
            queue->dry_at = sr >= n ? 0ull : n - sr;
        }
        if (lds) {
            if (idx32) VX_KL((k_walk<true, uint32_t>), grid, block, shmem, s, P); else VX_KL((k_walk<true, unsigned long long>), grid, block, shmem, s, P);
        } else if (wide) {
            if (idx32) VX_KL((k_walk<false, uint32_t, true>), grid, block, shmem, s, P); else VX_KL((k_walk<false, unsigned long long, true>), grid, block, shmem, s, P);
        } else {
            if (idx32) VX_KL((k_walk<false, uint32_t>), grid, block, shmem, s, P); else VX_KL((k_walk<false, unsigned long long>), grid, block, shmem, s, P);
        }
    }
}

}  // namespace vx
