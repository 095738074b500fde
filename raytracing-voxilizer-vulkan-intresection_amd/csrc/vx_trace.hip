// vx_trace.hip -- K6: first hit per ray against the occupied voxels' AABBs (replaces the procedural-hit stage
// raytrace.rint:46-71 that the reference runs under traceRayEXT, raytrace.rgen:49-64).
//
// The reference hands the occupied voxels' AABBs to the driver's BVH and runs raytrace.rint on every candidate; the result
// per ray is the minimum over ALL boxes of t0 = hitAabb(box) subject to t0 > 0 (rint:69) and tmin <= t0 <= tmax.  Here the
// occupancy itself is the acceleration structure: a 3-level 3D-DDA (64^3-cell blocks, 8^3-cell bricks, cells) enumerates a
// SUPERSET of the cells the ray can touch, and every occupied visited cell is put through the exact rint formula on the
// exact float box the reference would have built for it -- so the reported t is the very float the brute-force minimum
// yields.
//
// Data layout for traversal (built once per bitmask by k_build_bricks / k_brick_bounds / k_build_mip2, the analogue of the
// reference's BLAS build, hello_vulkan.cpp:737-760):
//   level 0  "bricks": the bitmask re-tiled brick-major -- one uint64 per (8x8x8 brick, z slice), bit = (y&7)*8 + (x&7), plus
//            a packed box of the brick's occupied cells.  A whole slice of cells costs ONE dependent 8-byte load instead of
//            one load per cell (the kernel is latency-bound: measured 47 % of wave cycles in s_waitcnt with the
//            reference-layout bitmask).
//   level 1  one bit per brick, x-fastest; staged in LDS by every workgroup when it fits (32 KiB at 512^3).
//   level 2  one bit per 8^3 bricks; a few hundred bytes, behind level 1 in LDS.
//
// Conservative enumeration.  A cell's float box differs from the nominal lattice planes by a few ulps of the largest
// coordinate.  Whenever two plane crossings are closer in t than that tolerance (per axis tau = tol_pos * |1/d|) the cells
// on the other side of the near-tie are LOOKED AT as well, forward (the other axes' next planes) and backward (planes just
// crossed).  Such neighbours are never walked -- the nominal ray does not pass through them -- only tested (level 0) or
// OR-ed into the "descend?" decision (upper levels); the walk always descends into the nominal cell and reaches the
// neighbours' children through the child level's own probes.  Traversal stops once the exit time of the current cell
// exceeds the best hit by more than the tolerance of the ray's major axis.
//
// Wave efficiency.  Persistent waves; a lane whose ray has finished takes the next ray of its wave's chunk of a global queue
// (first chunk static, later ones by one atomicAdd of a guided size).  A round alternates two phases so that the lanes run
// the same code together: upper-level walk steps (look at a cell / descend / advance / leave the block), then the brick
// test of every lane that posted an occupied brick (bit-parallel: slices x rows x candidate mask, exact slab test on the
// survivors).  Once the queue is dry, lanes with a long interval left hand its far half to idle lanes of their wave; the
// pieces meet in a 64-bit atomicMin per ray.  DESIGN.md section 4 has the measured history and what did not work.
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
// Brick-major re-tiling of the occupancy bitmask through LDS.  A workgroup takes 64 bricks in a row along x for one (by, bz):
// 64 voxel rows (8 z x 8 y) of 512 voxels.  Reads: row-contiguous words (a wave reads 64 consecutive bytes x 4 rows per
// instruction instead of 64 bytes in total), funnel-shifted to the chunk's own 32-voxel alignment.  Writes: thread t holds
// slice t%8 of brick t/8, i.e. consecutive threads write consecutive 8-byte words -- 4 KiB contiguous per workgroup.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_bricks(const uint32_t* __restrict__ words, uint32_t X, uint32_t Y, uint32_t Z, uint32_t BX, uint32_t BY,
                                                      uint32_t BZ, uint32_t chunks_x, uint64_t nwords, unsigned long long* __restrict__ bricks)
{
    __shared__ uint32_t rows[64][17];  // [z*8 + y][32-voxel chunk of the 512]; padded against bank conflicts of the column reads
    const uint64_t ngroups = (uint64_t)chunks_x * BY * BZ;
    for (uint64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint32_t cx = (uint32_t)(grp % chunks_x);
        const uint64_t q = grp / chunks_x;
        const uint32_t by = (uint32_t)(q % BY), bz = (uint32_t)(q / BY);
        const uint32_t x0 = cx * 512u;
        // ---- load: 64 rows x 16 words, four items per thread
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
        }
        __syncthreads();
        // ---- store: two (brick, slice) pairs per thread
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t item = (uint32_t)k * 256u + threadIdx.x;
            const uint32_t b = item >> 3, sl = item & 7u;
            const uint32_t bx = cx * 64u + b;
            if (bx < BX) {
                unsigned long long bits = 0ull;
                const uint32_t sh = (b & 3u) * 8u;
#pragma unroll
                for (uint32_t yy = 0; yy < 8u; ++yy) bits |= (unsigned long long)((rows[sl * 8u + yy][b >> 2] >> sh) & 0xFFu) << (yy * 8u);
                const uint64_t brick = (uint64_t)bx + (uint64_t)BX * ((uint64_t)by + (uint64_t)BY * bz);
                bricks[brick * 8ull + sl] = bits;
            }
        }
        __syncthreads();
    }
}

void launch_build_bricks(const uint32_t* words, const uint32_t dim[3], const uint32_t bdim[3], unsigned long long* bricks, hipStream_t s)
{
    const uint64_t n = (uint64_t)bdim[0] * bdim[1] * bdim[2];
    if (!n) return;
    const uint32_t chunks_x = (bdim[0] + 63u) / 64u;
    const uint64_t ngroups = (uint64_t)chunks_x * bdim[1] * bdim[2];
    const uint64_t nvox = (uint64_t)dim[0] * dim[1] * dim[2];
    const uint64_t nwords = (nvox + 31) / 32;
    uint64_t nblk = ngroups;
    if (nblk > 16384) nblk = 16384;
    VX_KL(k_build_bricks, dim3((unsigned)nblk), dim3(256), 0, s, words, dim[0], dim[1], dim[2], bdim[0], bdim[1], bdim[2], chunks_x, nwords, bricks);
}

// Per brick: the bounding box of its occupied cells, 3 bits per bound (xmin | xmax<<3 | ymin<<6 | ymax<<9 | zmin<<12 |
// zmax<<15).  Voxelized meshes are thin shells: a ray that crosses a wall's or a floor's brick without touching the
// one-voxel layer is rejected by one box-vs-box test instead of a walk over the brick's slices and rows.
__global__ __launch_bounds__(256) void k_brick_bounds(const unsigned long long* __restrict__ bricks, uint64_t nbricks, uint32_t* __restrict__ bounds,
                                                      uint32_t* __restrict__ m1)
{
    // one pass over whole 64-brick groups (so that the level-1 mip -- one bit per brick, 64 per wave -- is written as two
    // plain words per wave instead of being rebuilt from the bitmask by a second kernel)
    const uint64_t ngroups = (nbricks + 63) / 64;
    for (uint64_t gidx = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6; gidx < ngroups; gidx += ((uint64_t)gridDim.x * 256u) >> 6) {
        const uint64_t b = gidx * 64 + (threadIdx.x & 63);
        unsigned long long any = 0;
        uint32_t zmin = 7, zmax = 0;
        if (b < nbricks) {
            for (uint32_t s = 0; s < 8u; ++s) {
                const unsigned long long v = bricks[b * 8ull + s];
                if (v) { zmin = s < zmin ? s : zmin; zmax = s; }
                any |= v;
            }
        }
        uint32_t out = 0;
        if (any) {
            uint32_t rows = 0, cols = 0;  // rows: which y have a bit; cols: which x have a bit
            for (uint32_t y = 0; y < 8u; ++y) {
                const uint32_t byte = (uint32_t)(any >> (8u * y)) & 0xFFu;
                if (byte) rows |= 1u << y;
                cols |= byte;
            }
            const uint32_t xmin = __ffs(cols) - 1, xmax = 31 - __clz(cols), ymin = __ffs(rows) - 1, ymax = 31 - __clz(rows);
            out = xmin | (xmax << 3) | (ymin << 6) | (ymax << 9) | (zmin << 12) | (zmax << 15) | (1u << 18);
        }
        if (b < nbricks) bounds[b] = out;
        const unsigned long long occ = __ballot(any != 0);
        if ((threadIdx.x & 63) == 0) {
            m1[gidx * 2] = (uint32_t)occ;
            m1[gidx * 2 + 1] = (uint32_t)(occ >> 32);
        }
    }
}

void launch_brick_bounds(const unsigned long long* bricks, uint64_t nbricks, uint32_t* bounds, uint32_t* m1, hipStream_t s)
{
    if (!nbricks) return;
    uint64_t nblk = (nbricks + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    VX_KL(k_brick_bounds, dim3((unsigned)nblk), dim3(256), 0, s, bricks, nbricks, bounds, m1);
}

// Level-2 mip: one bit per 8x8x8 bricks.  One workgroup per OUTPUT WORD (32 blocks: 16 waves x 2), one lane per (y, z) row of
// a block's bricks: eight level-1 bits per lane, a ballot per block, one plain store per word -- no atomics, no memset.
__global__ __launch_bounds__(1024) void k_build_mip2(const uint32_t* __restrict__ m1, uint32_t d1x, uint32_t d1y, uint32_t d1z, uint32_t d2x, uint32_t d2y,
                                                     uint32_t d2z, uint32_t* __restrict__ m2)
{
    __shared__ uint32_t bits_s;
    if (threadIdx.x == 0) bits_s = 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t n2 = (uint64_t)d2x * d2y * d2z;
    uint32_t mine = 0u;
    for (uint32_t k = 0; k < 2u; ++k) {
        const uint32_t bit = wv * 2u + k;
        const uint64_t c = (uint64_t)blockIdx.x * 32u + bit;
        bool any = false;
        if (c < n2) {
            const uint32_t kz = (uint32_t)(c / ((uint64_t)d2x * d2y));
            const uint32_t rem = (uint32_t)(c - (uint64_t)kz * d2x * d2y);
            const uint32_t ky = rem / d2x, kx = rem - ky * d2x;
            const uint32_t by = ky * 8u + (lane & 7u), bz = kz * 8u + (lane >> 3), bx0 = kx * 8u;
            if (by < d1y && bz < d1z) {
                const uint32_t nb = d1x - bx0 < 8u ? d1x - bx0 : 8u;
                const uint64_t i0 = (uint64_t)bx0 + (uint64_t)d1x * ((uint64_t)by + (uint64_t)d1y * bz);
                const uint32_t sh = (uint32_t)i0 & 31u;
                uint32_t val = m1[i0 >> 5] >> sh;
                if (sh + nb > 32u) val |= m1[(i0 >> 5) + 1] << (32u - sh);
                any = (val & ((1u << nb) - 1u)) != 0u;
            }
        }
        if (__ballot(any)) mine |= 1u << bit;
    }
    if (lane == 0u && mine) atomicOr(&bits_s, mine);
    __syncthreads();
    if (threadIdx.x == 0) m2[blockIdx.x] = bits_s;
}

void launch_build_mip2(const uint32_t* m1, const uint32_t d1[3], const uint32_t d2[3], uint32_t* m2, hipStream_t s)
{
    const uint64_t n2 = (uint64_t)d2[0] * d2[1] * d2[2];
    if (!n2) return;
    VX_KL(k_build_mip2, dim3((unsigned)((n2 + 31) / 32)), dim3(1024), 0, s, m1, d1[0], d1[1], d1[2], d2[0], d2[1], d2[2], m2);
}

namespace {

// everything a lane carries for the ray it is currently tracing
struct Lane {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;  // origin, direction, 1/direction (rint:48)
    float taux, tauy, tauz;                    // crossing-time tolerance per axis
    float tn, tf;                              // entry / exit of the dilated grid box
    float best;                                // best accepted t so far
    uint64_t best_idx;                         // voxel index of the best hit
    // DDA state of the current level
    int cx, cy, cz;                            // current cell (in cells of this level)
    int px, py, pz;                            // parent cell (walk bounds of this level = its 8^3 children)
    int emask;                                 // axis through which the current cell was entered: 1 x, 2 y, 4 z, 0 = start cell
    float tau_ent;                             // tolerance of that axis (bit flags, not an axis index: an index makes the
                                               // compiler build scratch lookup tables out of the select chains)
    float tMx, tMy, tMz;                       // time of the next plane crossing per axis
    float tPx, tPy, tPz;                       // time of the plane behind per axis
    float t_in;                                // entry time of the current cell
    int lvl;
    float tolp;                                // position tolerance of this ray
    float tmax;                                // this ray's tMax (shadow rays: distance to the light)
    float tau_term;                            // termination margin: 2 x the tolerance of the ray's major axis
    // visit protocol of the current cell: cells still to look at (bit j = jx + 3*jy + 9*jz, 0 stay / 1 forward / 2 backward)
    uint32_t todo;
    bool fresh;                                // the current cell has not been expanded into `todo` yet
    bool occ;                                  // level 2: the block or a probed neighbour holds something
    bool pending;                              // level 1: an occupied brick waits for its brick test
    int bx, by, bz;                            // that brick
    uint32_t ob;                               // its occupied-cell bounds, loaded when the brick is posted
    uint32_t pf;                               // first word of its slices, loaded at the same time only to pull the line in
    // Exactly-zero direction components.  Such an axis has no plane crossings, so the near-tie probes above
    // never fire for it; instead the current cell's two neighbours along it are looked at whenever the (constant) coordinate
    // lies within the position tolerance of the cell's planes: znear bit 2a = upper neighbour, bit 2a+1 = lower neighbour.
    uint32_t znear;
};

// Axis selection BY VALUE.  `c ? R.x : R.y` on two struct members is an lvalue conditional: clang selects the ADDRESS and
// loads afterwards, which kept the lane state in scratch memory (dynamic scratch_load per step).  Passing the operands by
// value forces the loads first and the select stays in registers.
__device__ __forceinline__ float sel3(bool a, bool b, float x, float y, float z) { return a ? x : (b ? y : z); }
__device__ __forceinline__ int sel3(bool a, bool b, int x, int y, int z) { return a ? x : (b ? y : z); }

__device__ __forceinline__ float plane_t(float org, float vs, float o, float inv, int fine_index) { return ((org + (float)fine_index * vs) - o) * inv; }

// next / previous plane times of cell ci (cells of edge 1<<sh) along one axis
__device__ __forceinline__ void axis_planes(float& tM, float& tP, int ci, float o, float d, float inv, float org, float vs, int sh)
{
    const int scale = 1 << sh;
    const float a = plane_t(org, vs, o, inv, (ci + 1) * scale), b = plane_t(org, vs, o, inv, ci * scale);
    const bool pos = d >= 0.0f;
    tM = d == 0.0f ? INFINITY : (pos ? a : b);
    tP = d == 0.0f ? -INFINITY : (pos ? b : a);
}

__device__ __forceinline__ int start_cell(float o, float d, float org, float inv_vs, int sh, int lo, int hi, float t_lo)
{
    // the probes cover the start cell's rounding, so a reciprocal multiply is enough here
    const float p = o + t_lo * d;
    int ci = ((int)floorf((p - org) * inv_vs)) >> sh;
    ci = ci < lo ? lo : ci;
    return ci > hi - 1 ? hi - 1 : ci;
}

// zero-direction axis: is the constant coordinate o within tol of the upper (bit 0) / lower (bit 1) plane of cell ci?
__device__ __forceinline__ uint32_t zero_axis_near(float o, float org, float vs, int ci, int sh, float tol)
{
    const int scale = 1 << sh;
    const float lo = org + (float)(ci * scale) * vs, hi = org + (float)((ci + 1) * scale) * vs;
    return ((hi - o) <= tol ? 1u : 0u) | ((o - lo) <= tol ? 2u : 0u);
}
__device__ __forceinline__ void zero_axes_update(Lane& R, const GridParams& g, int sh)
{
    R.znear = 0u;
    if (R.dx == 0.0f || R.dy == 0.0f || R.dz == 0.0f) {  // rare: skipped by the whole wave unless one of its rays is axis-parallel
        const float tol = 2.0f * R.tolp;
        if (R.dx == 0.0f) R.znear |= zero_axis_near(R.ox, g.org[0], g.vs, R.cx, sh, tol);
        if (R.dy == 0.0f) R.znear |= zero_axis_near(R.oy, g.org[1], g.vs, R.cy, sh, tol) << 2;
        if (R.dz == 0.0f) R.znear |= zero_axis_near(R.oz, g.org[2], g.vs, R.cz, sh, tol) << 4;
    }
}

// enter level (sh = 3*level) inside walk bounds [lo, hi) at time t_lo
__device__ __forceinline__ void enter_level(Lane& R, const GridParams& g, float inv_vs, int sh, int lox, int loy, int loz, int hix, int hiy, int hiz,
                                            float t_lo)
{
    R.cx = start_cell(R.ox, R.dx, g.org[0], inv_vs, sh, lox, hix, t_lo);
    R.cy = start_cell(R.oy, R.dy, g.org[1], inv_vs, sh, loy, hiy, t_lo);
    R.cz = start_cell(R.oz, R.dz, g.org[2], inv_vs, sh, loz, hiz, t_lo);
    axis_planes(R.tMx, R.tPx, R.cx, R.ox, R.dx, R.ix, g.org[0], g.vs, sh);
    axis_planes(R.tMy, R.tPy, R.cy, R.oy, R.dy, R.iy, g.org[1], g.vs, sh);
    axis_planes(R.tMz, R.tPz, R.cz, R.oz, R.dz, R.iz, g.org[2], g.vs, sh);
    R.t_in = t_lo;
    R.emask = 0;
    R.tau_ent = 0.0f;
    zero_axes_update(R, g, sh);
}

// Ray r of the batch: from the ray buffer, or generated from the reference camera model (raytrace.rgen:41-47; mat*vec in glm's
// association (m0*v0 + m1*v1) + (m2*v2 + m3*v3)).
__device__ __forceinline__ void load_ray(bool primary, uint64_t r, const float* __restrict__ rays, const Camera* __restrict__ camp, float& ox, float& oy,
                                         float& oz, float& dx, float& dy, float& dz)
{
    if (primary) {
        const Camera& cam = *camp;  // in device memory: 34 dwords of kernel arguments would otherwise sit in (spilled) SGPRs
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

// Ray set-up: tolerances, grid clip, top-level start.  Returns false when the ray cannot touch the grid.
__device__ __forceinline__ bool setup_ray(Lane& R, const GridParams& g, const TraceMips& M, float inv_vs, float tmax, float seg_t0, float seg_t1)
{
    R.ix = 1.0f / R.dx; R.iy = 1.0f / R.dy; R.iz = 1.0f / R.dz;  // rint:48
    const float hx = g.org[0] + (float)g.dim[0] * g.vs, hy = g.org[1] + (float)g.dim[1] * g.vs, hz = g.org[2] + (float)g.dim[2] * g.vs;
    float Mx = fmaxf(fmaxf(fabsf(R.ox), fabsf(R.oy)), fabsf(R.oz));
    Mx = fmaxf(Mx, fmaxf(fmaxf(fabsf(g.org[0]), fabsf(g.org[1])), fabsf(g.org[2])));
    Mx = fmaxf(Mx, fmaxf(fmaxf(fabsf(hx), fabsf(hy)), fabsf(hz)));
    // position tolerance 16 * 2^-24 * max|coordinate|: the box planes carry <= 3 roundings of grid-sized numbers, the slab
    // formula subtracts the (possibly far) origin and multiplies by a rounded reciprocal (relative 2^-23 of the distance
    // travelled), and this kernel's own plane times carry the same again
    const float tolp = Mx * 9.5367431640625e-07f;
    R.taux = R.dx == 0.0f ? 0.0f : tolp * fabsf(R.ix);
    R.tauy = R.dy == 0.0f ? 0.0f : tolp * fabsf(R.iy);
    R.tauz = R.dz == 0.0f ? 0.0f : tolp * fabsf(R.iz);
    float tn = 0.0f, tf = tmax;
    bool miss = false;
#define VX_CLIP(o, d, inv, lo, hi)                                                                   \
    {                                                                                                 \
        const float t1 = (((lo)-tolp) - (o)) * (inv), t2 = (((hi) + tolp) - (o)) * (inv);             \
        const bool z = (d) == 0.0f;                                                                   \
        miss |= z && (((o) < (lo)-tolp) || ((o) > (hi) + tolp));                                      \
        tn = fmaxf(tn, z ? -INFINITY : fminf(t1, t2));                                                \
        tf = fminf(tf, z ? INFINITY : fmaxf(t1, t2));                                                 \
    }
    VX_CLIP(R.ox, R.dx, R.ix, g.org[0], hx)
    VX_CLIP(R.oy, R.dy, R.iy, g.org[1], hy)
    VX_CLIP(R.oz, R.dz, R.iz, g.org[2], hz)
#undef VX_CLIP
    tf += R.taux + R.tauy + R.tauz;
    tn = fmaxf(tn, seg_t0);  // a ray segment (see k_trace): only the cells overlapping [seg_t0, seg_t1] are walked
    tf = fminf(tf, seg_t1);
    R.tn = tn;
    R.tf = tf;
    R.best = INFINITY;
    R.best_idx = ~0ull;
    R.lvl = 2;
    R.tolp = tolp;
    R.tau_term = 2.0f * tolp / fmaxf(fmaxf(fabsf(R.dx), fabsf(R.dy)), fabsf(R.dz));
    R.todo = 0u;
    R.fresh = true;
    R.occ = false;
    R.pending = false;
    R.bx = R.by = R.bz = 0;
    R.ob = 0u;
    R.pf = 0u;
    R.px = R.py = R.pz = 0;
    R.znear = 0u;
    if (miss || !(tn <= tf) || !g.nvox) return false;
    // one virtual cell of halo around the top level: a ray sliding along the outside of a boundary face within tolerance
    // still walks next to the boundary cells and probes into them
    enter_level(R, g, inv_vs, 6, -1, -1, -1, (int)M.d2[0] + 1, (int)M.d2[1] + 1, (int)M.d2[2] + 1, tn);
    return true;
}

// Level 0: all cells of one 8x8x8 brick the (tolerance-dilated) ray can touch, bit-parallel.
// The brick is walked as z slices (one uint64 of occupancy each) and, inside a slice, as rows of 8 cells (one byte): for a
// row the ray's x range is turned into a bit mask and ANDed with the occupancy byte, and only the surviving bits go through
// the exact rint formula.  Every range is dilated by the position tolerance, so the tested set is a superset of the cells
// the reference's float boxes could report; the slab formula is the arbiter.  A ray skimming along an occupied wall for a
// whole brick costs a few row tests here instead of ~15 generic DDA steps (the tail of the step histogram: 1203 steps).
#ifdef VX_TRACE_DEBUG_UTIL
#define VX_BT_COUNT(i) ++bt_cnt[i];
#else
#define VX_BT_COUNT(i)
#endif
__device__ __forceinline__ void brick_test(Lane& R, const GridParams& g, const TraceMips& M, float inv_vs, float tolp, int bx, int by, int bz, float tmin,
                                           float tmax
#ifdef VX_TRACE_DEBUG_UTIL
                                           , int* bt_cnt  // [0] slices entered, [1] rows tested, [2] exact slab tests, [3] calls past the culling
#endif
)
{
    const float vs = g.vs;
    const int fx = bx * 8, fy = by * 8, fz = bz * 8;
    const float lox = g.org[0] + (float)fx * vs, loy = g.org[1] + (float)fy * vs, loz = g.org[2] + (float)fz * vs;
    const float hix = g.org[0] + (float)(fx + 8) * vs, hiy = g.org[1] + (float)(fy + 8) * vs, hiz = g.org[2] + (float)(fz + 8) * vs;
    const float tauS = R.taux + R.tauy + R.tauz;
    float ta = R.tn, tb = fminf(R.tf, R.best + tauS);
    // the ray inside the dilated brick
    // branch-free: a zero direction component makes the slab unbounded in t when the origin is inside it, empty otherwise
#define VX_CLIPB(o, d, inv, lo, hi)                                                                  \
    {                                                                                                 \
        const float t1 = (((lo)-tolp) - (o)) * (inv), t2 = (((hi) + tolp) - (o)) * (inv);             \
        const bool z = (d) == 0.0f, out = ((o) < (lo)-tolp) || ((o) > (hi) + tolp);                   \
        ta = fmaxf(ta, z ? (out ? INFINITY : -INFINITY) : fminf(t1, t2));                             \
        tb = fminf(tb, z ? INFINITY : fmaxf(t1, t2));                                                 \
    }
    VX_CLIPB(R.ox, R.dx, R.ix, lox, hix)
    VX_CLIPB(R.oy, R.dy, R.iy, loy, hiy)
    VX_CLIPB(R.oz, R.dz, R.iz, loz, hiz)
#undef VX_CLIPB
    if (!(ta <= tb)) return;
    const float tol2 = 2.0f * tolp;  // positions derived from a time carry the time's error as well
    const uint32_t bidx = (uint32_t)bx + M.d1[0] * ((uint32_t)by + M.d1[1] * (uint32_t)bz);
    // the (dilated) cell box of the ray segment inside the brick against the box of the brick's occupied cells
    const uint32_t ob = R.ob;
    const int oxmin = ob & 7u, oxmax = (ob >> 3) & 7u, oymin = (ob >> 6) & 7u, oymax = (ob >> 9) & 7u, ozmin = (ob >> 12) & 7u, ozmax = (ob >> 15) & 7u;
    const float xa0 = R.ox + ta * R.dx, xb0 = R.ox + tb * R.dx;
    const int cb0 = (int)floorf((fminf(xa0, xb0) - tol2 - lox) * inv_vs), cb1 = (int)floorf((fmaxf(xa0, xb0) + tol2 - lox) * inv_vs);
    if (cb1 < oxmin || cb0 > oxmax) return;
    const float ya0 = R.oy + ta * R.dy, yb0 = R.oy + tb * R.dy;
    const int rb0 = (int)floorf((fminf(ya0, yb0) - tol2 - loy) * inv_vs), rb1 = (int)floorf((fmaxf(ya0, yb0) + tol2 - loy) * inv_vs);
    if (rb1 < oymin || rb0 > oymax) return;
    const float za = R.oz + ta * R.dz, zb = R.oz + tb * R.dz;
    int s0 = (int)floorf((fminf(za, zb) - tol2 - loz) * inv_vs), s1 = (int)floorf((fmaxf(za, zb) + tol2 - loz) * inv_vs);
    s0 = s0 < ozmin ? ozmin : s0;
    s1 = s1 > ozmax ? ozmax : s1;
    if (s0 > s1) return;
    const unsigned long long* bp = M.bricks + (size_t)bidx * 8u;
    VX_BT_COUNT(3)
    const float o3[3] = {R.ox, R.oy, R.oz}, inv3[3] = {R.ix, R.iy, R.iz};
    const bool zfwd = R.dz >= 0.0f;
    for (int k = 0; k <= s1 - s0; ++k) {
        const int s = zfwd ? s0 + k : s1 - k;
        float tsa = ta, tsb = tb;
        {
            const float pl = g.org[2] + (float)(fz + s) * vs, ph = g.org[2] + (float)(fz + s + 1) * vs;
            const float t1 = ((pl - tolp) - R.oz) * R.iz, t2 = ((ph + tolp) - R.oz) * R.iz;
            const bool zd = R.dz == 0.0f;  // then the slice range [s0, s1] already is the set of slices the origin can be in
            tsa = fmaxf(tsa, zd ? -INFINITY : fminf(t1, t2));
            tsb = fminf(tsb, zd ? INFINITY : fmaxf(t1, t2));
        }
        if (!(tsa <= tsb)) continue;
        VX_BT_COUNT(0)
        const unsigned long long bits = bp[s];
        if (bits) {
            const float ya = R.oy + tsa * R.dy, yb = R.oy + tsb * R.dy;
            int r0 = (int)floorf((fminf(ya, yb) - tol2 - loy) * inv_vs), r1 = (int)floorf((fmaxf(ya, yb) + tol2 - loy) * inv_vs);
            r0 = r0 < oymin ? oymin : r0;
            r1 = r1 > oymax ? oymax : r1;
            // Candidates of the whole slice first, exact tests afterwards: nested divergent loops cost a wave the PRODUCT of its
            // lanes' longest trip counts (rows x candidates per row); two loops in sequence cost their sum.
            unsigned long long cand = 0ull;
            for (int r = r0; r <= r1; ++r) {
                const uint32_t rowbits = (uint32_t)(bits >> (8 * r)) & 0xFFu;
                if (!rowbits) continue;
                VX_BT_COUNT(1)
                float tra = tsa, trb = tsb;
                {
                    const float pl = g.org[1] + (float)(fy + r) * vs, ph = g.org[1] + (float)(fy + r + 1) * vs;
                    const float t1 = ((pl - tolp) - R.oy) * R.iy, t2 = ((ph + tolp) - R.oy) * R.iy;
                    const bool zd = R.dy == 0.0f;
                    tra = fmaxf(tra, zd ? -INFINITY : fminf(t1, t2));
                    trb = fminf(trb, zd ? INFINITY : fmaxf(t1, t2));
                }
                if (!(tra <= trb)) continue;
                const float xa = R.ox + tra * R.dx, xb = R.ox + trb * R.dx;
                int c0 = (int)floorf((fminf(xa, xb) - tol2 - lox) * inv_vs), c1 = (int)floorf((fmaxf(xa, xb) + tol2 - lox) * inv_vs);
                c0 = c0 < 0 ? 0 : c0;
                c1 = c1 > 7 ? 7 : c1;
                if (c0 > c1) continue;
                cand |= (unsigned long long)(rowbits & ((2u << c1) - (1u << c0))) << (8 * r);
            }
            while (cand) {
                const int b = __ffsll((long long)cand) - 1;
                cand &= cand - 1ull;
                VX_BT_COUNT(2)
                const uint32_t x = (uint32_t)(fx + (b & 7)), y = (uint32_t)(fy + (b >> 3)), z = (uint32_t)(fz + s);
                float bb[6];
                cell_aabb(g, x, y, z, bb);
                const float t = hit_aabb(bb, o3, inv3);                       // rint:46-56
                const uint64_t i = (uint64_t)x + (uint64_t)g.dim[0] * ((uint64_t)y + (uint64_t)g.dim[1] * (uint64_t)z);
                if (t > 0.0f && t >= tmin && t <= tmax &&                     // rint:69, rgen:50-51
                    (t < R.best || (t == R.best && i < R.best_idx))) {
                    R.best = t;
                    R.best_idx = i;
                }
            }
        }
        // cells of later slices are entered (in z) no earlier than this slice's dilated exit minus the z tolerance
        // (dz == 0: every candidate slice spans the same t range, a hit in one says nothing about the others)
        if (R.dz != 0.0f && R.best + 2.0f * R.tauz + R.tau_term < tsb) return;
    }
}

// One step of the upper-level walk (level 2: 64^3-cell blocks, level 1: bricks).  A step either looks at ONE cell of the
// current visit list (the walk's cell, then its near-tie neighbours) or finishes the cell (descend / advance / pop).  At
// level 1 an occupied cell is not processed here: it is posted as the lane's pending brick, and the caller runs the brick
// test for all lanes of the wave together (phase separation keeps the wave's lanes in the same code).
// Returns false when the ray is finished.
template <bool LDS_M1>
__device__ __forceinline__ bool upper_step(Lane& R, const GridParams& g, const TraceMips& M, const uint32_t* __restrict__ m1_lds, uint32_t m2_off, float inv_vs)
{
    const int lvl = R.lvl;
    const bool ex = (R.tMx <= R.tMy) && (R.tMx <= R.tMz);
    const bool ey = !ex && (R.tMy <= R.tMz);
    const float t_o = sel3(ex, ey, R.tMx, R.tMy, R.tMz);
    const int sx = R.dx < 0.0f ? -1 : 1, sy = R.dy < 0.0f ? -1 : 1, sz = R.dz < 0.0f ? -1 : 1;
    if (R.fresh) {
        // forward near-ties (other axes' next planes), backward near-ties (planes just behind) -> per-axis offset sets
        const float tau_exit = sel3(ex, ey, R.taux, R.tauy, R.tauz);
        const bool start = R.emask == 0;
        const bool ez = !ex && !ey;
        const uint32_t zn = R.znear;  // zero-direction axes: neighbours by position, not by crossing time
        const bool fx = ((zn & 1u) != 0u) || (!ex && (R.tMx - t_o <= tau_exit + R.taux));
        const bool fy = ((zn & 4u) != 0u) || (!ey && (R.tMy - t_o <= tau_exit + R.tauy));
        const bool fz = ((zn & 16u) != 0u) || (!ez && (R.tMz - t_o <= tau_exit + R.tauz));
        const bool bx = ((zn & 2u) != 0u) || (!(R.emask & 1) && (R.t_in - R.tPx <= R.taux + sel3(start, false, R.taux, 0.0f, R.tau_ent)));
        const bool by = ((zn & 8u) != 0u) || (!(R.emask & 2) && (R.t_in - R.tPy <= R.tauy + sel3(start, false, R.tauy, 0.0f, R.tau_ent)));
        const bool bz = ((zn & 32u) != 0u) || (!(R.emask & 4) && (R.t_in - R.tPz <= R.tauz + sel3(start, false, R.tauz, 0.0f, R.tau_ent)));
        const uint32_t mx = 1u | (fx ? 2u : 0u) | (bx ? 4u : 0u);
        const uint32_t my = 1u | (fy ? 1u << 3 : 0u) | (by ? 1u << 6 : 0u);
        const uint32_t mz = 1u | (fz ? 1u << 9 : 0u) | (bz ? 1u << 18 : 0u);
        R.todo = (mx * my) * mz;  // bit (jx + 3 jy + 9 jz) set iff every axis allows its offset; bit 0 = the cell itself
        R.fresh = false;
        R.occ = false;
    }
    if (R.todo) {
        const int j = __ffs(R.todo) - 1;
        R.todo &= R.todo - 1;
        const int jx = j % 3, jy = (j / 3) % 3, jz = j / 9;
        const int nx = R.cx + (jx == 0 ? 0 : (jx == 1 ? sx : -sx));
        const int ny = R.cy + (jy == 0 ? 0 : (jy == 1 ? sy : -sy));
        const int nz = R.cz + (jz == 0 ? 0 : (jz == 1 ? sz : -sz));
        if (lvl == 1) {
            if ((unsigned)nx < M.d1[0] && (unsigned)ny < M.d1[1] && (unsigned)nz < M.d1[2]) {
                const uint32_t i = (uint32_t)nx + M.d1[0] * ((uint32_t)ny + M.d1[1] * (uint32_t)nz);
                const uint32_t w = LDS_M1 ? m1_lds[i >> 5] : M.w1[i >> 5];
                if ((w >> (i & 31u)) & 1u) {
                    // post the brick; with nothing else to look at in this cell the walk moves on in the same step (the
                    // brick test runs before the lane's next step, and testing a brick late only delays the ray's end)
                    R.pending = true;
                    R.bx = nx; R.by = ny; R.bz = nz;
                    // start the brick's two dependent loads now: incoherent rays miss L2 on nearly every brick (16 MiB of
                    // slices at 512^3), and with four waves per SIMD that latency is not hidden inside the brick phase
                    R.ob = M.bounds[i];
                    R.pf = reinterpret_cast<const uint32_t*>(M.bricks)[(size_t)i * 16u];
                }
            }
        } else {
            if ((unsigned)nx < M.d2[0] && (unsigned)ny < M.d2[1] && (unsigned)nz < M.d2[2]) {
                const uint32_t i = (uint32_t)nx + M.d2[0] * ((uint32_t)ny + M.d2[1] * (uint32_t)nz);
                const uint32_t w = LDS_M1 ? m1_lds[m2_off + (i >> 5)] : M.w2[i >> 5];  // with the L1 mip in LDS the (tiny) L2 mip sits behind it
                R.occ |= ((w >> (i & 31u)) & 1u) != 0u;
            }
        }
        if (R.todo) return true;  // more cells to look at; otherwise finish the cell right away
    }
    // ---- finish the cell: descend (level 2, something found), terminate, advance, or leave the block (level 1).
    // All three moves end in "new cell at some level": its six plane times are recomputed from the cell index (plane_t is a
    // pure function of the index, so the recomputed values are the ones an incremental update would carry).  One shared tail
    // instead of three code paths: the wave pays for the union of its lanes' paths at every step.
    const bool desc = lvl == 2 && R.occ;
    // Termination.  Every cell not looked at yet has slab t0 >= t_o - tau(major axis): along the major axis the ray is
    // monotone and well conditioned, so cells of later major-axis slabs are entered no earlier than t_o - tau_major; cells of
    // the current slab that are reached through another axis were either flagged as near-ties and looked at just now, or
    // their crossing is more than the tolerance away.  (The sum of all three taus is NOT needed: one tiny direction
    // component would make it infinite and force those rays through the whole grid -- the tail of the kernel.)
    if (!desc && !(t_o <= fminf(R.tf, R.best + R.tau_term))) return false;
    int ncx = R.cx, ncy = R.cy, ncz = R.cz, nl = lvl;
    float t_new = t_o;
    if (desc) {
        // into the NOMINAL block's bricks, starting exactly at its entry time (a time slack would slide the start point
        // along the ray's major axis; the start cell's rounding is covered by the brick walk's probes)
        R.px = R.cx; R.py = R.cy; R.pz = R.cz;
        t_new = fmaxf(R.t_in, R.tn);
        const int lx = R.cx * 8, ly = R.cy * 8, lz = R.cz * 8;
        ncx = start_cell(R.ox, R.dx, g.org[0], inv_vs, 3, lx, lx + 8, t_new);
        ncy = start_cell(R.oy, R.dy, g.org[1], inv_vs, 3, ly, ly + 8, t_new);
        ncz = start_cell(R.oz, R.dz, g.org[2], inv_vs, 3, lz, lz + 8, t_new);
        nl = 1;
    } else {
        const int s_a = sel3(ex, ey, sx, sy, sz);
        const int c_a = sel3(ex, ey, R.cx, R.cy, R.cz) + s_a;
        const int p_a = sel3(ex, ey, R.px, R.py, R.pz);
        const int top_hi = sel3(ex, ey, (int)M.d2[0], (int)M.d2[1], (int)M.d2[2]) + 1;
        const int lo_a = lvl == 2 ? -1 : p_a * 8;
        const int hi_a = lvl == 2 ? top_hi : p_a * 8 + 8;
        if (c_a < lo_a || c_a >= hi_a) {
            if (lvl == 2) return false;  // left the grid (and its halo)
            // Left the block -- through a face that is also the BLOCK's exit face: the plane is the same fine index at both
            // levels, so the block's own exit time is this t_o and its exit axis this axis (the other axes' block planes lie at
            // or behind the brick planes, and ties resolve by the same x, y, z priority).  So the block advances right here,
            // instead of popping to the block level and spending a step on "visited, advance".
            const int q_a = p_a + s_a;
            if (q_a < -1 || q_a >= top_hi) return false;
            ncx = ex ? q_a : R.px;
            ncy = ey ? q_a : R.py;
            ncz = (!ex && !ey) ? q_a : R.pz;
            nl = 2;
        } else {
            ncx = ex ? c_a : ncx;
            ncy = ey ? c_a : ncy;
            ncz = (!ex && !ey) ? c_a : ncz;
        }
    }
    const int sh = nl * 3;
    R.cx = ncx; R.cy = ncy; R.cz = ncz;
    R.lvl = nl;
    axis_planes(R.tMx, R.tPx, ncx, R.ox, R.dx, R.ix, g.org[0], g.vs, sh);
    axis_planes(R.tMy, R.tPy, ncy, R.oy, R.dy, R.iy, g.org[1], g.vs, sh);
    axis_planes(R.tMz, R.tPz, ncz, R.oz, R.dz, R.iz, g.org[2], g.vs, sh);
    R.t_in = t_new;
    R.emask = desc ? 0 : (ex ? 1 : (ey ? 2 : 4));
    R.tau_ent = desc ? 0.0f : sel3(ex, ey, R.taux, R.tauy, R.tauz);
    R.fresh = true;
    R.todo = 0u;
    R.occ = false;
    zero_axes_update(R, g, sh);
    return true;
}

}  // namespace

// position of the n-th (0-based) set bit of a 64-bit mask (n < popcount)
__device__ __forceinline__ int nth_set_bit64(unsigned long long m, int n)
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

__device__ __forceinline__ float shfl_f(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src)
{
    return ((unsigned long long)__shfl((unsigned)(v >> 32), src, 64) << 32) | __shfl((unsigned)v, src, 64);
}

__device__ __forceinline__ unsigned long long pack_key(float t, unsigned long long idx) { return ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)idx; }

// Everything the kernel is given, as ONE by-value argument.  The walk needs ~25 uniform values in scalar registers at every
// step; the other ~35 dwords (ray source, outputs, work counters) are touched once per refill / retire.  Held as ordinary
// kernel arguments they all stay live in SGPRs for the whole kernel and are spilled to VGPR lanes and read back
// (v_readlane: 16 % of the kernel's vector instructions).  The cold block is therefore read from the kernarg segment where
// it is used (scalar loads through a laundered pointer, which the compiler cannot hoist out of the loop).
struct TraceHot {
    GridParams g;
    TraceMips M;
    float tmin;
    int any_hit;
    uint32_t m1_words;
    uint32_t m2_words;
};
struct TraceCold {
    const float* rays;            // null: primary rays from *cam
    const Camera* cam;
    const float* tmax_per_ray;    // null: tmax
    float tmax;
    uint32_t pad;
    uint64_t nrays;
    float* t_out;
    unsigned long long* idx_out;
    uint8_t* shadowed_out;
    unsigned long long* next_item;   // work counter
    unsigned long long* keys;        // per ray: atomicMin merge key of the pieces of a split ray
    uint8_t* split_flag;             // per ray: 1 once the ray has been split (null: no work donation); all zero between launches
};
struct TraceParams {
    TraceHot hot;
    TraceCold cold;
};

typedef const TraceCold __attribute__((address_space(4)))* ColdPtr;
__device__ __forceinline__ ColdPtr cold_params()
{
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (ColdPtr)(p + offsetof(TraceParams, cold));
}

#ifdef VX_TRACE_DEBUG_UTIL
// diagnostic build: where the wave cycles go.  [0..4] cycles in refill / donate / walk / brick test / retire,
// [5] walk iterations, [6] sum of active lanes over them, [7] brick phases, [8] sum of lanes with a pending brick,
// [9] rounds, [10] sum of busy lanes at round start
__device__ unsigned long long g_trace_util[24];
#define VX_UTIL_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); dbg_c[i] += now_ - dbg_last; dbg_last = now_; }
#define VX_UTIL_ADD(i, v) dbg_c[i] += (unsigned long long)(v);
#else
#define VX_UTIL_T(i)
#define VX_UTIL_ADD(i, v)
#endif

// Persistent waves with dynamic work fetch.  Exit condition every wave reaches: the work counter passes the ray count (no
// refill possible) and every lane's ray (or piece of a ray) has finished; each finishes in a bounded number of steps, and a
// piece is only split while its t interval is longer than six bricks.
#ifndef VX_T_STEPS
#define VX_T_STEPS 4
#endif
#ifndef VX_T_ITERS
#define VX_T_ITERS 2
#endif
#ifndef VX_T_CHUNK
#define VX_T_CHUNK 64
#endif
#ifndef VX_T_CHUNK_MAX
#define VX_T_CHUNK_MAX 256
#endif
#ifndef VX_T_REFILL
#define VX_T_REFILL 44
#endif
template <bool LDS_M1>
__global__ __launch_bounds__(256, 4 /*waves per SIMD: keeps the allocation at <= 128 VGPRs*/) void k_trace(const TraceParams P)
{
    constexpr int kStepsPerRound = VX_T_STEPS;   // upper-level steps between two brick-test phases
    constexpr int kItersPerRound = VX_T_ITERS;   // (walk, brick test) iterations between two refill checks
    constexpr int kRefillBelow = VX_T_REFILL;    // refill when fewer than this many lanes are busy
    constexpr int kChunkRays = VX_T_CHUNK;  // rays a wave reserves per touch of the global counter: at least ...
    constexpr int kChunkMax = VX_T_CHUNK_MAX;  // ... and at most
    constexpr int kDonateBelow = 48;    // drain phase: donate work while at most this many lanes are busy
    constexpr float kDonateBricks = 6.0f;  // ... and only from pieces with more than this many bricks of t interval left
    // (swept on the MI355X in round 1: the kernel is insensitive to all of them within +-5 %)
    const GridParams& g = P.hot.g;
    const TraceMips& M = P.hot.M;
    extern __shared__ __attribute__((aligned(16))) uint32_t m1_lds[];
    if (LDS_M1) {
        for (uint32_t i = threadIdx.x; i < P.hot.m1_words; i += 256u) m1_lds[i] = M.w1[i];
        for (uint32_t i = threadIdx.x; i < P.hot.m2_words; i += 256u) m1_lds[P.hot.m1_words + i] = M.w2[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const float inv_vs = 1.0f / g.vs;
    Lane R;
    uint64_t r = ~0ull;      // ray this lane is tracing (~0: none)
    bool busy = false;       // traversal in progress
    bool shared = false;     // this lane walks a PIECE of a split ray: its result goes through the ray's atomicMin key
    bool drained = false;    // no ray left for this wave: the global counter and the wave's chunk are exhausted
    bool drained_global = false;
    uint64_t chunk_cur = 0, chunk_end = 0;
#ifdef VX_TRACE_DEBUG_STEPS
    int dbg_steps = 0;
#endif
#ifdef VX_TRACE_DEBUG_UTIL
    unsigned long long dbg_c[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int bt_tot[4] = {0, 0, 0, 0};
    int bt_work = 0;
    unsigned long long dbg_last = __builtin_readcyclecounter();
    unsigned long long dbg_drain_at = ~0ull;
#endif
#ifdef VX_TRACE_DEBUG_CYCLES
    unsigned long long dbg_t0 = 0;
#endif
    // The first chunk of every wave is assigned statically: thousands of waves asking the one counter in the same microsecond
    // queue up behind each other at the memory-side atomic unit.
    uint64_t static_rays;
    {
        const ColdPtr C = cold_params();
        const uint64_t nrays = C->nrays;
        uint64_t csz = nrays / (8ull * gridDim.x);
        csz = csz > (uint64_t)kChunkMax ? (uint64_t)kChunkMax : csz;
        csz = csz < (uint64_t)kChunkRays ? (uint64_t)kChunkRays : (csz & ~63ull);
        static_rays = csz * 4ull * gridDim.x;
        chunk_cur = csz * (4ull * blockIdx.x + (threadIdx.x >> 6));
        chunk_end = chunk_cur + csz;
        if (chunk_end > nrays) chunk_end = nrays;
        if (chunk_cur > chunk_end) chunk_cur = chunk_end;
        if (static_rays >= nrays) drained_global = true;
    }
    for (;;) {
        const unsigned long long busy_mask = __ballot(busy);
        const int nbusy = __popcll(busy_mask);
        if (!drained && nbusy < kRefillBelow) {
            // ---- refill idle lanes.  Ray indices come from a per-wave chunk; the global counter is touched once per chunk.
            const ColdPtr C = cold_params();
            const uint64_t nrays = C->nrays;
            const unsigned long long idle_mask = ~busy_mask;
            const uint64_t need = (uint64_t)(64 - nbusy);
            const uint64_t take = need < chunk_end - chunk_cur ? need : chunk_end - chunk_cur;
            const uint64_t first = chunk_cur;
            chunk_cur += take;
            uint64_t second = 0;
            if (take < need) {
                // Guided chunk size: the wave waits ~2 us for the counter's old value, so it asks for a large chunk while much is
                // left (half an even share of what remained at its previous fetch) and for the minimum near the end, where
                // balance matters (fixed sizes: 32 -> 0.63 ms, 64 -> 0.467, 128 -> 0.444, 256 -> 0.471 at 1M rays).
                const uint64_t left = nrays > chunk_end ? nrays - chunk_end : 0;
                uint64_t csz = left / (8ull * gridDim.x);  // 4 waves per workgroup, half a share
                csz = csz > (uint64_t)kChunkMax ? (uint64_t)kChunkMax : csz;
                csz = csz < (uint64_t)kChunkRays ? (uint64_t)kChunkRays : (csz & ~63ull);
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(C->next_item, (unsigned long long)csz);
                base = shfl_u64(base, 0) + static_rays;  // the counter runs behind the statically assigned first chunks
                second = base;
                chunk_cur = base + (need - take);
                chunk_end = base + csz;
                if (chunk_end > nrays) chunk_end = nrays > base ? nrays : base;
                if (chunk_cur > chunk_end) chunk_cur = chunk_end;
                if (base + csz >= nrays) drained_global = true;
            }
            if (drained_global && chunk_cur >= chunk_end) {
                drained = true;
#ifdef VX_TRACE_DEBUG_UTIL
                if (dbg_drain_at == ~0ull) { dbg_drain_at = 0; for (int i = 0; i < 5; ++i) dbg_drain_at += dbg_c[i]; }
#endif
            }
            if (!busy) {
                const uint64_t pos = (uint64_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                const uint64_t mine = pos < take ? first + pos : (take < need ? second + (pos - take) : nrays);
                if (mine < nrays) {
                    r = mine;
                    shared = false;
                    const float* rays = C->rays;
                    load_ray(rays == nullptr, r, rays, C->cam, R.ox, R.oy, R.oz, R.dx, R.dy, R.dz);
                    const float* tpr = C->tmax_per_ray;
                    const float tmax_r = tpr ? tpr[r] : C->tmax;
                    R.tmax = tmax_r;
                    busy = setup_ray(R, g, M, inv_vs, tmax_r, 0.0f, INFINITY);
#ifdef VX_TRACE_DEBUG_CYCLES
                    dbg_t0 = wall_clock64();
#endif
#ifdef VX_TRACE_DEBUG_STEPS
                    dbg_steps = 0;
#endif
                    if (!busy) {  // cannot touch the grid: retire at once
                        float* t_out = C->t_out;
                        unsigned long long* idx_out = C->idx_out;
                        uint8_t* shadowed_out = C->shadowed_out;
                        if (t_out) t_out[r] = -1.0f;
                        if (idx_out) idx_out[r] = ~0ull;
                        if (shadowed_out) shadowed_out[r] = 0;
                        r = ~0ull;
                    }
                }
            }
        }
        VX_UTIL_T(0)
        const unsigned long long bm = __ballot(busy);
        if (!bm) {
            if (drained) break;
            continue;
        }
        VX_UTIL_ADD(9, 1)
        VX_UTIL_ADD(10, __popcll(bm))
        // ---- work donation (drain phase).  With 4 rays per lane at 1M rays the queue empties early and every wave is left
        // with a few dozen long rays on a shrinking set of lanes.  Once no new ray can be fetched, a busy lane with enough of
        // its t interval left hands the FAR half to an idle lane of its wave (ray and interval travel by shuffles); the two
        // pieces are walked independently (every cell overlapping the piece, with the usual probes) and meet in a 64-bit
        // atomicMin of (t bits << 32 | voxel index) -- the same "closest, then lower index" order a single walk uses.
        // The outputs of the rays that were split are written from their keys afterwards (k_rank, or k_merge_flags).
        if (drained && __popcll(bm) <= kDonateBelow) {
            const ColdPtr C = cold_params();
            uint8_t* split_flag = C->split_flag;
            if (split_flag) {
                const int nb = __popcll(bm);
                const float t_cur = fmaxf(R.t_in, R.tn), t_end = fminf(R.tf, R.best + R.tau_term);
                const float brick_time = 8.0f * g.vs / fmaxf(fmaxf(fabsf(R.dx), fabsf(R.dy)), fabsf(R.dz));
                const bool can = busy && !R.pending && (t_end - t_cur > kDonateBricks * brick_time);
                const unsigned long long dm = __ballot(can);
                const unsigned long long im = ~bm;
                const int ndon = min(__popcll(dm), 64 - nb);
                if (ndon > 0) {
                    // donor side: am I among the first ndon donors?
                    const int drank = __popcll(dm & ((1ull << lane) - 1ull));
                    const bool donor = can && drank < ndon;
                    const float t_mid = 0.5f * (t_cur + t_end);
                    // a split ray's outputs are written from its merge key afterwards (k_merge_flags / k_rank): mark it
                    if (donor) split_flag[r] = 1;
                    // receiver side: the q-th idle lane takes the q-th donor's far half
                    const int irank = __popcll(im & ((1ull << lane) - 1ull));
                    const bool recv = !busy && irank < ndon;
                    const int src = nth_set_bit64(dm, recv ? irank : 0);
                    const float s_ox = shfl_f(R.ox, src), s_oy = shfl_f(R.oy, src), s_oz = shfl_f(R.oz, src);
                    const float s_dx = shfl_f(R.dx, src), s_dy = shfl_f(R.dy, src), s_dz = shfl_f(R.dz, src);
                    const float s_tmax = shfl_f(R.tmax, src), s_mid = shfl_f(t_mid, src), s_end = shfl_f(t_end, src), s_best = shfl_f(R.best, src);
                    const unsigned long long s_bidx = shfl_u64(R.best_idx, src), s_r = shfl_u64(r, src);
                    if (donor) {
                        R.tf = t_mid;  // keep the near half (the walk ends with the cell that contains t_mid)
                        shared = true;
                    }
                    if (recv) {
                        r = s_r;
                        R.ox = s_ox; R.oy = s_oy; R.oz = s_oz; R.dx = s_dx; R.dy = s_dy; R.dz = s_dz;
                        R.tmax = s_tmax;
                        busy = setup_ray(R, g, M, inv_vs, s_tmax, s_mid, s_end);
                        R.best = s_best;
                        R.best_idx = s_bidx;
                        shared = true;
                        if (!busy) r = ~0ull;
                    }
                }
            }
        }
        // ---- trace, in two phases so that the wave's lanes run the same code together:
        // (1) upper-level walk until the lane has an occupied brick pending (or its ray is finished),
        // (2) the brick test for every lane with a pending brick.
        bool finished = false;
        VX_UTIL_T(1)
        for (int it = 0; it < kItersPerRound; ++it) {
            for (int k = 0; k < kStepsPerRound; ++k) {
                const bool go = busy && !finished && !R.pending;
                const unsigned long long gm = __ballot(go);
                if (!gm) break;
                VX_UTIL_ADD(5, 1)
                VX_UTIL_ADD(6, __popcll(gm))
                if (go) {
                    if (!upper_step<LDS_M1>(R, g, M, m1_lds, P.hot.m1_words, inv_vs)) finished = true;
#ifdef VX_TRACE_DEBUG_STEPS
#if VX_TRACE_DEBUG_STEPS == 3
                    {   // count only steps whose cell lies outside the grid (halo block / halo bricks)
                        const bool outside = R.lvl == 2 ? ((unsigned)R.cx >= M.d2[0] || (unsigned)R.cy >= M.d2[1] || (unsigned)R.cz >= M.d2[2])
                                                        : ((unsigned)R.cx >= M.d1[0] || (unsigned)R.cy >= M.d1[1] || (unsigned)R.cz >= M.d1[2]);
                        if (outside) ++dbg_steps;
                    }
#elif VX_TRACE_DEBUG_STEPS != 2
                    ++dbg_steps;
#endif
#endif
                }
            }
            const bool pend = busy && R.pending;
            const unsigned long long pm = __ballot(pend);
            VX_UTIL_T(2)
            if (!pm) break;  // every live lane finished its ray in this round
            VX_UTIL_ADD(7, 1)
            VX_UTIL_ADD(8, __popcll(pm))
            if (pend) {
                asm volatile("" ::"v"(R.pf));  // the prefetching load has to be "used"
#ifdef VX_TRACE_DEBUG_UTIL
                int bt_cnt[4] = {0, 0, 0, 0};
                brick_test(R, g, M, inv_vs, R.tolp, R.bx, R.by, R.bz, P.hot.tmin, R.tmax, bt_cnt);
                bt_tot[0] += bt_cnt[0]; bt_tot[1] += bt_cnt[1]; bt_tot[2] += bt_cnt[2]; bt_tot[3] += bt_cnt[3];
                bt_work = bt_cnt[0] * 20 + bt_cnt[1] * 40 + bt_cnt[2] * 60;
#else
                brick_test(R, g, M, inv_vs, R.tolp, R.bx, R.by, R.bz, P.hot.tmin, R.tmax);
#endif
#if defined(VX_TRACE_DEBUG_STEPS) && VX_TRACE_DEBUG_STEPS == 2
                ++dbg_steps;  // diagnostic: brick tests per ray
#endif
                R.pending = false;
                // shadow query (gl_RayFlagsTerminateOnFirstHitEXT, raytrace2.rchit:108): any accepted hit ends the ray
                if (P.hot.any_hit && R.best_idx != ~0ull) finished = true;
                // the walk already stands in the next cell (entered at t_in = the exit time of the brick's cell): apply the
                // walk's termination rule with the new best now instead of spending a step on it
                if (R.fresh && R.emask != 0 && !(R.t_in <= fminf(R.tf, R.best + R.tau_term))) finished = true;
            }
#ifdef VX_TRACE_DEBUG_UTIL
            {   // divergence of the brick phase: the wave pays the slowest lane's work
                int mx = pend ? bt_work : 0, sm = pend ? bt_work : 0;
                for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(mx, m, 64); mx = o > mx ? o : mx; sm += __shfl_xor(sm, m, 64); }
                dbg_c[11] += (unsigned long long)mx;
                dbg_c[12] += (unsigned long long)sm;
            }
#endif
            VX_UTIL_T(3)
        }
        VX_UTIL_T(2)
        // ---- retire: t and the voxel index of the hit; the primitive rank (two dependent loads), the normal and the hit
        // compaction are done by k_rank over all rays afterwards, off this kernel's critical path
        if (__ballot(finished)) {
            const ColdPtr C = cold_params();
            if (finished) {
                if (shared) {
                    if (R.best_idx != ~0ull) atomicMin(C->keys + r, pack_key(R.best, R.best_idx));
                } else {
                    float best_t = R.best_idx != ~0ull ? R.best : -1.0f;
#ifdef VX_TRACE_DEBUG_CYCLES
                    best_t = (float)(wall_clock64() - dbg_t0);  // diagnostic build: report the ray's residency in 100 MHz ticks
#endif
#ifdef VX_TRACE_DEBUG_STEPS
                    best_t = (float)dbg_steps;  // diagnostic build: report the step count instead of t
#endif
                    float* t_out = C->t_out;
                    unsigned long long* idx_out = C->idx_out;
                    uint8_t* shadowed_out = C->shadowed_out;
                    if (t_out) t_out[r] = best_t;
                    if (idx_out) idx_out[r] = R.best_idx;
                    if (shadowed_out) shadowed_out[r] = R.best_idx != ~0ull ? 1 : 0;
                }
                busy = false;
                r = ~0ull;
            }
        }
        VX_UTIL_T(4)
    }
#ifdef VX_TRACE_DEBUG_UTIL
    for (int i = 0; i < 4; ++i) {
        int v = bt_tot[i];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) atomicAdd(&g_trace_util[16 + i], (unsigned long long)v);
    }
    if (lane == 0) {
        unsigned long long life = 0;
        for (int i = 0; i < 13; ++i) { atomicAdd(&g_trace_util[i], dbg_c[i]); if (i < 5) life += dbg_c[i]; }
        atomicMax(&g_trace_util[20], life);                                  // longest wave
        atomicAdd(&g_trace_util[21], life > dbg_drain_at ? life - dbg_drain_at : 0ull);  // cycles after the wave's queue ran dry
        atomicAdd(&g_trace_util[22], 1ull);                                  // waves
    }
#endif
}

#ifdef VX_TRACE_DEBUG_UTIL
}  // namespace vx
extern "C" int vx_debug_trace_util(unsigned long long* out24, int reset)
{
    if (out24 && hipMemcpyFromSymbol(out24, HIP_SYMBOL(vx::g_trace_util), 24 * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[24] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(vx::g_trace_util), z, 24 * 8) != hipSuccess) return 1; }
    return 0;
}
namespace vx {
#endif

// Outputs of the rays that were split by work donation, from their merged keys.  The flags are read four rays at a time and
// cleared on the way (they must be all zero for the next launch).  Used when k_rank does not run; k_rank does the same itself.
__global__ __launch_bounds__(256) void k_merge_flags(uint8_t* __restrict__ flags, uint64_t nrays, const unsigned long long* __restrict__ keys,
                                                     float* __restrict__ t_out, unsigned long long* __restrict__ idx_out, uint8_t* __restrict__ shadowed_out)
{
    const uint64_t nquads = (nrays + 3) / 4;
    uint32_t* f4 = reinterpret_cast<uint32_t*>(flags);  // the buffer is padded to a multiple of 4 bytes
    for (uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x; q < nquads; q += (uint64_t)gridDim.x * 256u) {
        const uint32_t w = f4[q];
        if (!w) continue;
        f4[q] = 0u;
        for (uint32_t k = 0; k < 4u; ++k) {
            if (!((w >> (8u * k)) & 0xFFu)) continue;
            const uint64_t r = q * 4u + k;
            if (r >= nrays) break;
            const unsigned long long key = keys[r];
            if (t_out) t_out[r] = key == ~0ull ? -1.0f : __uint_as_float((uint32_t)(key >> 32));
            if (idx_out) idx_out[r] = key == ~0ull ? ~0ull : (key & 0xFFFFFFFFull);
            if (shadowed_out) shadowed_out[r] = key == ~0ull ? 0 : 1;
        }
    }
}

// Per-ray post-pass over all rays: primitive id (== gl_PrimitiveID: rank of the voxel in the ascending AABB list), the
// cube-face normal of raytrace2.rchit:60-73, and wavefront hit compaction.
__global__ __launch_bounds__(1024) void k_rank(float* __restrict__ t, void* __restrict__ idx_any, int idx32, uint64_t nrays, GridParams g,
                                              const uint32_t* __restrict__ words, const uint32_t* __restrict__ word_prefix, const float* __restrict__ rays,
                                              const Camera* __restrict__ cam, uint32_t* __restrict__ prim_out, float* __restrict__ normal_out, vx_hit* __restrict__ hits,
                                              unsigned long long* nhits, uint8_t* __restrict__ split_flag, const unsigned long long* __restrict__ keys,
                                              uint8_t* __restrict__ shadowed_out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = r < nrays;
    float tt = -1.0f;
    uint32_t prim = 0xFFFFFFFFu;
    if (active) {
        tt = t[r];
        unsigned long long* idx = reinterpret_cast<unsigned long long*>(idx_any);  // k_trace writes 64-bit voxel indices, k_walk 32-bit ones when they fit
        unsigned long long i;
        if (idx32) { const uint32_t i32 = reinterpret_cast<const uint32_t*>(idx_any)[r]; i = i32 == 0xFFFFFFFFu ? ~0ull : (unsigned long long)i32; }
        else i = idx[r];
        if (split_flag && split_flag[r]) {  // a ray that was split by work donation: its result is the merge key (and the flag goes back to 0)
            const unsigned long long key = keys[r];
            tt = key == ~0ull ? -1.0f : __uint_as_float((uint32_t)(key >> 32));
            i = key == ~0ull ? ~0ull : (key & 0xFFFFFFFFull);
            t[r] = tt;
            idx[r] = i;
            if (shadowed_out) shadowed_out[r] = key == ~0ull ? 0 : 1;
            split_flag[r] = 0;
        }
        float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
        if (i != ~0ull) {
            const uint64_t wi = i >> 5;
            const uint32_t bit = (uint32_t)i & 31u;
            prim = word_prefix[wi] + __popc(words[wi] & ((1u << bit) - 1u));
            if (normal_out) {
                const uint64_t XY = (uint64_t)g.dim[0] * g.dim[1];
                const uint32_t z = (uint32_t)(i / XY);
                const uint32_t rem = (uint32_t)(i - (uint64_t)z * XY);
                const uint32_t y = rem / g.dim[0], x = rem - y * g.dim[0];
                float bb[6], ox, oy, oz, dx, dy, dz;
                cell_aabb(g, x, y, z, bb);
                load_ray(rays == nullptr, r, rays, cam, ox, oy, oz, dx, dy, dz);
                // worldPos = origin + direction * t; worldNrm = normalize(worldPos - (min + max) * 0.5)      rchit:60-65
                const float vx_ = (ox + dx * tt) - ((bb[0] + bb[3]) * 0.5f);
                const float vy_ = (oy + dy * tt) - ((bb[1] + bb[4]) * 0.5f);
                const float vz_ = (oz + dz * tt) - ((bb[2] + bb[5]) * 0.5f);
                const float il = 1.0f / sqrtf((vx_ * vx_ + vy_ * vy_) + vz_ * vz_);
                const float nx = vx_ * il, ny = vy_ * il, nz = vz_ * il;
                const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
                const float maxC = fmaxf(fmaxf(ax, ay), az);                                                // rchit:70
                if (maxC == ax) n0 = nx > 0.0f ? 1.0f : (nx < 0.0f ? -1.0f : 0.0f);                         // rchit:71-73
                else if (maxC == ay) n1 = ny > 0.0f ? 1.0f : (ny < 0.0f ? -1.0f : 0.0f);
                else n2 = nz > 0.0f ? 1.0f : (nz < 0.0f ? -1.0f : 0.0f);
            }
        }
        if (prim_out) prim_out[r] = prim;
        if (normal_out) { normal_out[3 * r] = n0; normal_out[3 * r + 1] = n1; normal_out[3 * r + 2] = n2; }
    }
    if (hits) {
        // Compaction: ONE touch of the global counter per workgroup (launched with 1024 threads for this).  One per wave meant
        // 15 600 returning atomics on one address for 1M rays, ~10 ns each at the memory-side atomic unit: 196 us instead of 23.
        __shared__ unsigned wcnt[16];
        __shared__ unsigned long long bbase;
        const bool hit = active && prim != 0xFFFFFFFFu;
        const unsigned long long bal = __ballot(hit);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
        if (lane == 0) wcnt[wv] = (unsigned)__popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned tot = 0;
            for (int w = 0; w < nw; ++w) tot += wcnt[w];
            bbase = tot ? atomicAdd(nhits, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        if (hit) {
            unsigned long long off = bbase;
            for (int w = 0; w < wv; ++w) off += wcnt[w];
            vx_hit h;
            h.ray = (uint32_t)r; h.prim = prim; h.t = tt;
            hits[off + __popcll(bal & ((1ull << lane) - 1ull))] = h;
        }
    }
}

size_t trace_spill_bytes(uint64_t nrays) { return (size_t)nrays + 64; }  // one flag byte per ray: split by work donation

void launch_walk(const GridParams& g, const TraceMips& mips, const unsigned long long* bricks3, const TraceIO& io, unsigned long long* counter,
                 void* idx_out, bool idx32, hipStream_t s);

bool trace_uses_walk()
{
    const char* e = getenv("VOXHIP_TRACE_ALGO");  // "dda": the round-1 three-level DDA with near-tie probes (kept for A/B); default: slab walk
    return !(e && std::strcmp(e, "dda") == 0);
}

void launch_trace(const GridParams& g, const TraceMips& mips, const unsigned long long* bricks3, const uint32_t* word_prefix, const TraceIO& io,
                  unsigned long long* counters /*>= 4*/, unsigned long long* idx_tmp, void* spill_buf, unsigned long long* keys, hipStream_t s)
{
    const uint64_t nrays = io.nrays;
    if (!nrays) return;
    if (trace_uses_walk()) {
        const bool want_rank = (io.prim_out || io.hits || io.normal_out) && word_prefix && idx_tmp && io.t_out;
        const bool idx32 = g.nvox < 0xFFFFFFFFull;
        launch_walk(g, mips, bricks3, io, counters, want_rank ? (void*)idx_tmp : nullptr, idx32, s);
        if (want_rank) {
            if (io.hits && io.nhits) (void)hipMemsetAsync(io.nhits, 0, sizeof(unsigned long long), s);
            const unsigned rthreads = io.hits ? 1024u : 256u;  // the hit list's compaction touches the global counter once per workgroup
            const dim3 rgrid((unsigned)((nrays + rthreads - 1) / rthreads)), rblock(rthreads);
            VX_KL(k_rank, rgrid, rblock, 0, s, io.t_out, (void*)idx_tmp, idx32 ? 1 : 0, nrays, g, mips.w0, word_prefix, io.rays, io.cam_dev, io.prim_out, io.normal_out,
                  io.hits, io.nhits, (uint8_t*)nullptr, (const unsigned long long*)nullptr, io.shadowed_out);
        }
        return;
    }
    // counters[0]: work counter
    (void)hipMemsetAsync(counters, 0, sizeof(unsigned long long), s);
    const uint64_t n1 = (uint64_t)mips.d1[0] * mips.d1[1] * mips.d1[2];
    const uint32_t m1_words = (uint32_t)((n1 + 31) / 32);
    // VOXHIP_TRACE_LDS=0 forces the global-memory mips (the path every grid above ~550^3 takes) -- used by the parity tests
    const char* env_lds = getenv("VOXHIP_TRACE_LDS");
    const bool lds_m1 = (size_t)m1_words * 4 <= 40960 && !(env_lds && atoi(env_lds) == 0);  // 4 workgroups x 40 KiB fit the CU's 160 KiB
    // persistent grid: 256 CUs x 4 resident 256-thread workgroups, fewer when there are not that many rays
    // Small batches (a few rays per lane) are bound by the drain of their longest rays and run faster on three waves per SIMD
    // than on four (1M rays: 0.50 ms at 768 workgroups, 0.54 ms at 1024; equal at 2M; 8M: 2.32 vs 2.12 ms).
    const int env_blocks = getenv("VOXHIP_TRACE_BLOCKS") ? atoi(getenv("VOXHIP_TRACE_BLOCKS")) : 0;
    const uint64_t max_blocks = env_blocks > 0 ? (uint64_t)env_blocks : (nrays <= 1500000ull ? 768ull : 1024ull);
    uint64_t nblk = (nrays + 255) / 256;
    if (nblk > max_blocks) nblk = max_blocks;
    const dim3 grid((unsigned)nblk), block(256);
    const bool want_rank = (io.prim_out || io.hits || io.normal_out) && word_prefix && idx_tmp && io.t_out;
    unsigned long long* idx_out = want_rank ? idx_tmp : nullptr;
    const uint64_t n2 = (uint64_t)mips.d2[0] * mips.d2[1] * mips.d2[2];
    const uint32_t m2_words = (uint32_t)((n2 + 31) / 32);
    const size_t shmem = lds_m1 ? (size_t)(m1_words + m2_words) * 4 : 0;
    // intra-wave work donation in the drain phase (default on; VOXHIP_TRACE_DONATE=0 disables).  The merge key holds the
    // voxel index in 32 bits.  spill_buf: one flag byte per ray, all zero on entry (the caller zeroes a new buffer) and on exit.
    const int env_donate = getenv("VOXHIP_TRACE_DONATE") ? atoi(getenv("VOXHIP_TRACE_DONATE")) : 1;
    const bool donate = env_donate && spill_buf && keys && g.nvox <= 0x100000000ull && nrays < 0xFFFFFFFFull;
    uint8_t* flags = donate ? (uint8_t*)spill_buf : nullptr;
    if (donate) (void)hipMemsetAsync(keys, 0xFF, (size_t)nrays * 8, s);  // "no hit" in every ray's merge key
    TraceParams P;
    std::memset(&P, 0, sizeof(P));
    P.hot.g = g;
    P.hot.M = mips;
    P.hot.tmin = io.tmin;
    P.hot.any_hit = io.any_hit ? 1 : 0;
    P.hot.m1_words = m1_words;
    P.hot.m2_words = m2_words;
    P.cold.rays = io.rays;
    P.cold.cam = io.cam_dev;  // device copy of the camera (null for explicit rays)
    P.cold.tmax_per_ray = io.tmax_per_ray;
    P.cold.tmax = io.tmax;
    P.cold.nrays = nrays;
    P.cold.t_out = io.t_out;
    P.cold.idx_out = idx_out;
    P.cold.shadowed_out = io.shadowed_out;
    P.cold.next_item = counters;
    P.cold.keys = keys;
    P.cold.split_flag = flags;
    if (lds_m1) { VX_KL(k_trace<true>, grid, block, shmem, s, P); } else { VX_KL(k_trace<false>, grid, block, shmem, s, P); }
    if (want_rank) {
        if (io.hits && io.nhits) (void)hipMemsetAsync(io.nhits, 0, sizeof(unsigned long long), s);
        const unsigned rthreads = io.hits ? 1024u : 256u;  // the hit list's compaction touches the global counter once per workgroup
        const dim3 rgrid((unsigned)((nrays + rthreads - 1) / rthreads)), rblock(rthreads);
        VX_KL(k_rank, rgrid, rblock, 0, s, io.t_out, (void*)idx_tmp, 0, nrays, g, mips.w0, word_prefix, io.rays, io.cam_dev, io.prim_out, io.normal_out, io.hits, io.nhits,
              flags, keys, io.shadowed_out);
    } else if (donate) {
        const uint64_t nq = (nrays + 3) / 4;
        VX_KL(k_merge_flags, dim3((unsigned)((nq + 255) / 256 > 4096 ? 4096 : (nq + 255) / 256)), block, 0, s, flags, nrays, keys, io.t_out, idx_out, io.shadowed_out);
    }
}

}  // namespace vx
