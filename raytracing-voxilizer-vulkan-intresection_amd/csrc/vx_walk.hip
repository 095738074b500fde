// vx_walk.hip -- K6: first hit per ray against the occupied voxels' AABBs (replaces the procedural-hit stage raytrace.rint:46-71
// that the reference runs under traceRayEXT, raytrace.rgen:49-64) -- major-axis slab walk.
//
// The reference hands the occupied voxels' AABBs to the driver's BVH and runs raytrace.rint on every candidate; the result per
// ray is the minimum over ALL boxes of t0 = hitAabb(box) subject to t0 > 0 (rint:69) and tmin <= t0 <= tmax.  Here the occupancy
// itself is the acceleration structure, and the enumeration of candidate cells is organised so that it is cheap, uniform across
// the lanes of a wave, and provably a superset of the cells whose float boxes the ray can enter:
//
//   Let w be the axis of the largest |direction| component, u and v the other two.  The grid is cut into slabs perpendicular to
//   w at three granularities: 64 cells (blocks, level 2), 8 cells (bricks, level 1), 1 cell (level 0).  For a slab with lattice
//   planes P_near, P_far (in travel order) the ray is within the position tolerance `tol` of the slab for
//        t in [ta, tb],  ta = inv_w * ((P_near -/+ tol) - o_w),  tb = inv_w * ((P_far +/- tol) - o_w)
//   -- the same expression form as hitAabb's `invDir * (plane - origin)` (rint:49-50), so by monotonicity of float subtraction
//   and multiplication every box of the slab has a COMPUTED entry time >= ta: once the best accepted t is < ta of a slab, no box
//   of that slab or of any later one can beat it (exact, no slack needed).  Inside [ta, tb] the ray's u and v coordinates stay in
//        [min(p(ta), p(tb)) - 2 tol, max(p(ta), p(tb)) + 2 tol]
//   (|d_u|, |d_v| <= |d_w|: an error in t moves the point by less than the same error along w; no 1/d_u blow-up exists, and a
//   zero component simply gives a constant coordinate), so the cells the ray can touch in the slab lie in that rectangle --
//   1x1 .. 2x2 cells of the slab's granularity, 3x3 at worst.  A step of the walk = one slab: its rectangle is looked up in the
//   occupancy mip of its level; an occupied rectangle descends to the eight finer slabs, an empty one moves on.  At level 0 the
//   rectangle's occupied cells go through the exact rint formula on the float box the reference would have built for them,
//   which is the only arbiter of hit and t -- the reported t is the very float the brute-force minimum yields.
//
// Data layout (built once per bitmask by k_build_bricks3 / k_brick_bounds / k_build_mip2, the analogue of the reference's BLAS
// build, hello_vulkan.cpp:737-760):
//   level 0  "bricks": the bitmask re-tiled brick-major in THREE orientations, one per possible major axis: for orientation w one
//            uint64 per (8x8x8 brick, slab along w), bit = (v&7)*8 + (u&7) with (u, v) = the axes after w cyclically.  Whatever
//            the ray's major axis, the cells of a 1-cell slab inside a brick are ONE 8-byte load.
//   level 1  one bit per brick, x-fastest; level 2 one bit per 8^3 bricks; both staged in LDS by every workgroup when they fit.
//
// Wave efficiency.  Persistent waves; a lane whose ray has finished takes the next ray of its wave's chunk of a global queue
// (first chunk static, later ones by one atomicAdd of a guided size).  Every active lane executes the same slab step whatever
// its level; only the exact tests (1.7 per ray on the bench scene) diverge.
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

__global__ __launch_bounds__(256) void k_build_bricks3(const uint32_t* __restrict__ words, uint32_t X, uint32_t Y, uint32_t Z, uint32_t BX, uint32_t BY,
                                                       uint32_t BZ, uint32_t chunks_x, uint64_t nwords, unsigned long long* __restrict__ bricks3,
                                                       uint64_t ori_stride /*uint64 words per orientation*/)
{
    __shared__ uint32_t rows[64][17];              // [z*8 + y][32-voxel chunk of the 512]; padded against bank conflicts of the column reads
    __shared__ unsigned long long sz[64][9];       // [brick][z slab]: bit y*8 + x (padded)
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
        // ---- z orientation: two (brick, slab) pairs per thread; kept in LDS for the transposes
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t item = (uint32_t)k * 256u + threadIdx.x;
            const uint32_t b = item >> 3, sl = item & 7u;
            unsigned long long bits = 0ull;
            const uint32_t sh = (b & 3u) * 8u;
#pragma unroll
            for (uint32_t yy = 0; yy < 8u; ++yy) bits |= (unsigned long long)((rows[sl * 8u + yy][b >> 2] >> sh) & 0xFFu) << (yy * 8u);
            sz[b][sl] = bits;
            const uint32_t bx = cx * 64u + b;
            if (bx < BX) {
                const uint64_t brick = (uint64_t)bx + (uint64_t)BX * ((uint64_t)by + (uint64_t)BY * bz);
                bricks3[2ull * ori_stride + brick * 8ull + sl] = bits;
            }
        }
        __syncthreads();
        // ---- x and y orientations: thread = (brick, orientation, half of the slabs)
        {
            const uint32_t b = threadIdx.x & 63u, part = threadIdx.x >> 6;  // part 0,1: x slabs 0-3 / 4-7; part 2,3: y slabs 0-3 / 4-7
            const uint32_t bx = cx * 64u + b;
            unsigned long long s[8];
            unsigned long long any = 0ull;
#pragma unroll
            for (int z = 0; z < 8; ++z) { s[z] = sz[b][z]; any |= s[z]; }
            if (bx < BX) {
                const uint64_t brick = (uint64_t)bx + (uint64_t)BX * ((uint64_t)by + (uint64_t)BY * bz);
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
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j] = out[j];
            }
        }
        __syncthreads();
    }
}

void launch_build_bricks3(const uint32_t* words, const uint32_t dim[3], const uint32_t bdim[3], unsigned long long* bricks3, hipStream_t s)
{
    const uint64_t n = (uint64_t)bdim[0] * bdim[1] * bdim[2];
    if (!n) return;
    const uint32_t chunks_x = (bdim[0] + 63u) / 64u;
    const uint64_t ngroups = (uint64_t)chunks_x * bdim[1] * bdim[2];
    const uint64_t nvox = (uint64_t)dim[0] * dim[1] * dim[2];
    const uint64_t nwords = (nvox + 31) / 32;
    uint64_t nblk = ngroups;
    if (nblk > 16384) nblk = 16384;
    VX_KL(k_build_bricks3, dim3((unsigned)nblk), dim3(256), 0, s, words, dim[0], dim[1], dim[2], bdim[0], bdim[1], bdim[2], chunks_x, nwords, bricks3, n * 8ull);
}

namespace {

// everything a lane carries for the ray it is currently tracing
struct WalkLane {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;  // origin, direction, 1/direction (rint:48)
    float ou, du, ov, dv, ow, iw;              // the same permuted: w = major axis, u, v = the axes after it cyclically
    float orgu, orgv, orgw;                    // grid origin, permuted
    int dimu, dimv, dimw;                      // grid dims in cells, permuted
    int perm;                                  // w: 0 x, 1 y, 2 z
    float tol;                                 // position tolerance
    float tn, tf;                              // the ray inside the dilated grid box, cut to [0, tmax]
    float tmax;
    float best;                                // best accepted t so far (+inf: none)
    uint64_t best_idx;                         // voxel index of the best hit
    int lvl, k;                                // current slab: level (2 blocks, 1 bricks, 0 cells) and index along w in cells of that level
};

__device__ __forceinline__ float sel3f(int p, float a, float b, float c) { return p == 0 ? a : (p == 1 ? b : c); }
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
__device__ __forceinline__ bool walk_setup(WalkLane& R, const GridParams& g, const uint32_t d2[3], float inv_vs, float tmax)
{
    R.ix = 1.0f / R.dx; R.iy = 1.0f / R.dy; R.iz = 1.0f / R.dz;  // rint:48
    const float ax = fabsf(R.dx), ay = fabsf(R.dy), az = fabsf(R.dz);
    const int p = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    R.perm = p;
    R.ow = sel3f(p, R.ox, R.oy, R.oz); R.iw = sel3f(p, R.ix, R.iy, R.iz);
    R.ou = sel3f(p, R.oy, R.oz, R.ox); R.du = sel3f(p, R.dy, R.dz, R.dx);
    R.ov = sel3f(p, R.oz, R.ox, R.oy); R.dv = sel3f(p, R.dz, R.dx, R.dy);
    R.orgw = sel3f(p, g.org[0], g.org[1], g.org[2]);
    R.orgu = sel3f(p, g.org[1], g.org[2], g.org[0]);
    R.orgv = sel3f(p, g.org[2], g.org[0], g.org[1]);
    R.dimw = sel3i(p, (int)g.dim[0], (int)g.dim[1], (int)g.dim[2]);
    R.dimu = sel3i(p, (int)g.dim[1], (int)g.dim[2], (int)g.dim[0]);
    R.dimv = sel3i(p, (int)g.dim[2], (int)g.dim[0], (int)g.dim[1]);
    const float hx = g.org[0] + (float)g.dim[0] * g.vs, hy = g.org[1] + (float)g.dim[1] * g.vs, hz = g.org[2] + (float)g.dim[2] * g.vs;
    float Mx = fmaxf(fmaxf(fabsf(R.ox), fabsf(R.oy)), fabsf(R.oz));
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
    VX_CLIP(R.ox, R.dx, R.ix, g.org[0], hx)
    VX_CLIP(R.oy, R.dy, R.iy, g.org[1], hy)
    VX_CLIP(R.oz, R.dz, R.iz, g.org[2], hz)
#undef VX_CLIP
    // the clip's own rounding: widen by the time it takes to travel 2 tol along the major axis
    const float tslack = 2.0f * tol * fabsf(R.iw);
    tn = fmaxf(tn - tslack, 0.0f);
    tf = tf + tslack;
    R.tn = tn;
    R.tf = tf;
    R.tmax = tmax;
    R.best = INFINITY;
    R.best_idx = ~0ull;
    R.lvl = 2;
    R.k = 0;
    if (miss || !(tn <= tf)) return false;
    // first block slab: the one that holds the entry point (one cell of slack against the rounding of this estimate; slabs that
    // still lie in front of tn are skipped by the step itself)
    const bool pos = R.iw > 0.0f;
    const float pw = sel3f(p, R.ox + tn * R.dx, R.oy + tn * R.dy, R.oz + tn * R.dz);
    int cw = (int)floorf((pw - R.orgw) * inv_vs) + (pos ? -1 : 1);
    cw = cw < 0 ? 0 : (cw > R.dimw - 1 ? R.dimw - 1 : cw);
    R.k = cw >> 6;
    return true;
}

}  // namespace

struct WalkParams {
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
    // touched once per refill / retire
    const float* rays;            // null: primary rays from *cam
    const Camera* cam;
    const float* tmax_per_ray;    // null: tmax
    float tmax;
    uint32_t idx32;               // 1: idx_out holds 32-bit voxel indices (grids of at most 2^32 - 1 voxels), 0xFFFFFFFF = miss
    uint64_t nrays;
    float* t_out;
    void* idx_out;
    uint8_t* shadowed_out;
    unsigned long long* next_item;   // work counter
};

#ifndef VX_W_STEPS
#define VX_W_STEPS 6
#endif
#ifndef VX_W_REFILL
#define VX_W_REFILL 48
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
#define VX_W_MINWAVES 1
#endif

// One slab of the walk.  Returns false when the ray is finished.
template <bool LDS_MIPS>
__device__ __forceinline__ bool walk_step(WalkLane& R, const WalkParams& P, const uint32_t* __restrict__ mips_lds)
{
    const GridParams& g = P.g;
    const int lvl = R.lvl, sh = 3 * lvl, k = R.k;
    const bool pos = R.iw > 0.0f;
    // ---- the slab's [ta, tb] along the major axis
    const int i0 = k << sh;
    int i1 = (k + 1) << sh;
    i1 = i1 > R.dimw ? R.dimw : i1;
    const float lo = (R.orgw + (float)i0 * g.vs) - R.tol, hi = (R.orgw + (float)i1 * g.vs) + R.tol;
    const float t_lo = R.iw * (lo - R.ow), t_hi = R.iw * (hi - R.ow);
    const float ta = pos ? t_lo : t_hi, tb = pos ? t_hi : t_lo;
    if (R.best < ta || ta > R.tf) return false;  // no box of this slab or any later one can beat the best hit / beyond the interval
    const float ca = fmaxf(ta, R.tn), cb = fminf(tb, R.tf);
    bool any = false;
    if (ca <= cb) {
        // ---- the rectangle of cells (of this level) the ray can touch inside the slab
        const float tol2 = 2.0f * R.tol;
        const float ua = R.ou + ca * R.du, ub = R.ou + cb * R.du;
        const float va = R.ov + ca * R.dv, vb = R.ov + cb * R.dv;
        int u0 = (int)floorf(((fminf(ua, ub) - tol2) - R.orgu) * P.inv_vs), u1 = (int)floorf(((fmaxf(ua, ub) + tol2) - R.orgu) * P.inv_vs);
        int v0 = (int)floorf(((fminf(va, vb) - tol2) - R.orgv) * P.inv_vs), v1 = (int)floorf(((fmaxf(va, vb) + tol2) - R.orgv) * P.inv_vs);
        u0 = u0 < 0 ? 0 : u0;
        v0 = v0 < 0 ? 0 : v0;
        u1 = u1 > R.dimu - 1 ? R.dimu - 1 : u1;
        v1 = v1 > R.dimv - 1 ? R.dimv - 1 : v1;
        // candidates are looked up in a mip: blocks at level 2, bricks at levels 1 AND 0 (a level-0 slab reads the slab words of the
        // occupied bricks its rectangle touches)
        const int shc = lvl == 0 ? 3 : sh;
        const int cu0 = u0 >> shc, cu1 = u1 >> shc, cv0 = v0 >> shc, cv1 = v1 >> shc;  // (u1, v1 may be -1: arithmetic shift keeps them negative)
        const int kc = lvl == 0 ? (k >> 3) : k;
        const bool top = lvl == 2;
        const uint32_t Dx = top ? P.d2[0] : P.d1[0], Dy = top ? P.d2[1] : P.d1[1];
        const uint32_t moff = top ? P.m1_words : 0u;
        for (int cv = cv0; cv <= cv1; ++cv) {
            for (int cu = cu0; cu <= cu1; ++cu) {
                int x, y, z;
                unperm(R.perm, cu, cv, kc, x, y, z);
                const uint32_t i = (uint32_t)x + Dx * ((uint32_t)y + Dy * (uint32_t)z);
                const uint32_t wd = LDS_MIPS ? mips_lds[moff + (i >> 5)] : (top ? P.w2[i >> 5] : P.w1[i >> 5]);
                if (!((wd >> (i & 31u)) & 1u)) continue;
                if (lvl != 0) { any = true; continue; }
                // ---- level 0: the slab's word of this brick, the rectangle as a bit mask, exact tests on what survives
                const unsigned long long bits = P.bricks3[(uint64_t)R.perm * P.ori_stride + (uint64_t)i * 8ull + (uint32_t)(k & 7)];
                const int bu = cu << 3, bv = cv << 3;
                const int a0 = (u0 > bu ? u0 : bu) - bu, a1 = (u1 < bu + 7 ? u1 : bu + 7) - bu;  // columns inside the brick
                const int b0 = (v0 > bv ? v0 : bv) - bv, b1 = (v1 < bv + 7 ? v1 : bv + 7) - bv;  // rows
                const unsigned long long col = (unsigned long long)((2u << a1) - (1u << a0)) * 0x0101010101010101ull;
                const unsigned long long rowsel = (~0ull >> (8 * (7 - b1))) & (~0ull << (8 * b0));
                unsigned long long cand = bits & col & rowsel;
                while (cand) {
                    const int b = __ffsll((long long)cand) - 1;
                    cand &= cand - 1ull;
                    int cx, cy, cz;
                    unperm(R.perm, bu + (b & 7), bv + (b >> 3), k, cx, cy, cz);
                    float bb[6];
                    cell_aabb(g, (uint32_t)cx, (uint32_t)cy, (uint32_t)cz, bb);
                    const float o3[3] = {R.ox, R.oy, R.oz}, inv3[3] = {R.ix, R.iy, R.iz};
                    const float t = hit_aabb(bb, o3, inv3);                       // rint:46-56
                    const uint64_t vi = (uint64_t)(uint32_t)cx + (uint64_t)g.dim[0] * ((uint64_t)(uint32_t)cy + (uint64_t)g.dim[1] * (uint64_t)(uint32_t)cz);
                    if (t > 0.0f && t >= P.tmin && t <= R.tmax &&                 // rint:69, rgen:50-51
                        (t < R.best || (t == R.best && vi < R.best_idx))) {
                        R.best = t;
                        R.best_idx = vi;
                    }
                }
            }
        }
        if (P.any_hit && R.best_idx != ~0ull) return false;  // shadow query (gl_RayFlagsTerminateOnFirstHitEXT, raytrace2.rchit:108)
    }
    // ---- next slab: down into the eight finer slabs of an occupied rectangle, else on (and up when the parent's slabs are done)
    if (any) {
        const int nl = lvl - 1;
        const int ncw = (R.dimw + (1 << (3 * nl)) - 1) >> (3 * nl);
        int kf = pos ? (k << 3) : (k << 3) + 7;
        kf = kf > ncw - 1 ? ncw - 1 : kf;
        R.lvl = nl;
        R.k = kf;
        return true;
    }
    const int s = pos ? 1 : -1;
    int nl = lvl, kk = k, kn = k + s;
    if (nl < 2 && (kn >> 3) != (kk >> 3)) { ++nl; kk >>= 3; kn = kk + s; }
    if (nl < 2 && (kn >> 3) != (kk >> 3)) { ++nl; kk >>= 3; kn = kk + s; }
    const int ncw = (R.dimw + (1 << (3 * nl)) - 1) >> (3 * nl);
    if (kn < 0 || kn >= ncw) return false;  // left the grid
    R.lvl = nl;
    R.k = kn;
    return true;
}

// Persistent waves with dynamic work fetch.  Exit condition every wave reaches: the work counter passes the ray count (no
// refill possible) and every lane's ray has finished; a ray finishes in a bounded number of slabs (its index along the major
// axis is monotone at every level).
template <bool LDS_MIPS>
__global__ __launch_bounds__(VX_W_BLOCK, VX_W_MINWAVES) void k_walk(const WalkParams P)
{
    constexpr int kStepsPerRound = VX_W_STEPS;   // slab steps between two refill checks
    constexpr int kRefillBelow = VX_W_REFILL;    // refill when fewer than this many lanes are busy
    constexpr int kChunkRays = VX_W_CHUNK;       // rays a wave reserves per touch of the global counter: at least ...
    constexpr int kChunkMax = VX_W_CHUNK_MAX;    // ... and at most
    constexpr unsigned kWavesPerBlock = VX_W_BLOCK / 64;
    const GridParams& g = P.g;
    extern __shared__ __attribute__((aligned(16))) uint32_t mips_lds[];
    if (LDS_MIPS) {
        for (uint32_t i = threadIdx.x; i < P.m1_words; i += VX_W_BLOCK) mips_lds[i] = P.w1[i];
        for (uint32_t i = threadIdx.x; i < P.m2_words; i += VX_W_BLOCK) mips_lds[P.m1_words + i] = P.w2[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    WalkLane R;
    uint64_t r = ~0ull;      // ray this lane is tracing (~0: none)
    bool busy = false;       // traversal in progress
    bool drained = false;    // no ray left for this wave: the global counter and the wave's chunk are exhausted
    bool drained_global = false;
    uint64_t chunk_cur = 0, chunk_end = 0;
    const uint64_t nrays = P.nrays;
    // The first chunk of every wave is assigned statically: thousands of waves asking the one counter in the same microsecond
    // queue up behind each other at the memory-side atomic unit.
    uint64_t static_rays;
    {
        uint64_t csz = nrays / (2ull * kWavesPerBlock * gridDim.x);
        csz = csz > (uint64_t)kChunkMax ? (uint64_t)kChunkMax : csz;
        csz = csz < (uint64_t)kChunkRays ? (uint64_t)kChunkRays : (csz & ~63ull);
        static_rays = csz * kWavesPerBlock * gridDim.x;
        chunk_cur = csz * ((uint64_t)kWavesPerBlock * blockIdx.x + (threadIdx.x >> 6));
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
            const unsigned long long idle_mask = ~busy_mask;
            const uint64_t need = (uint64_t)(64 - nbusy);
            const uint64_t take = need < chunk_end - chunk_cur ? need : chunk_end - chunk_cur;
            const uint64_t first = chunk_cur;
            chunk_cur += take;
            uint64_t second = 0;
            if (take < need && !drained_global) {
                // Guided chunk size: the wave waits ~2 us for the counter's old value, so it asks for a large chunk while much is
                // left (half an even share of what remained at its previous fetch) and for the minimum near the end.
                const uint64_t left = nrays > chunk_end ? nrays - chunk_end : 0;
                uint64_t csz = left / (2ull * kWavesPerBlock * gridDim.x);
                csz = csz > (uint64_t)kChunkMax ? (uint64_t)kChunkMax : csz;
                csz = csz < (uint64_t)kChunkRays ? (uint64_t)kChunkRays : (csz & ~63ull);
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(P.next_item, (unsigned long long)csz);
                base = (((unsigned long long)__shfl((unsigned)(base >> 32), 0, 64) << 32) | __shfl((unsigned)base, 0, 64)) + static_rays;
                second = base;
                chunk_cur = base + (need - take);
                chunk_end = base + csz;
                if (chunk_end > nrays) chunk_end = nrays > base ? nrays : base;
                if (chunk_cur > chunk_end) chunk_cur = chunk_end;
                if (base + csz >= nrays) drained_global = true;
            }
            if (drained_global && chunk_cur >= chunk_end) drained = true;
            if (!busy) {
                const uint64_t posn = (uint64_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                const uint64_t mine = posn < take ? first + posn : (second ? second + (posn - take) : nrays);
                if (mine < nrays && (posn < take || mine < chunk_end)) {
                    r = mine;
                    load_ray_w(P.rays == nullptr, r, P.rays, P.cam, R.ox, R.oy, R.oz, R.dx, R.dy, R.dz);
                    const float tmax_r = P.tmax_per_ray ? P.tmax_per_ray[r] : P.tmax;
                    busy = walk_setup(R, g, P.d2, P.inv_vs, tmax_r);
                    if (!busy) {  // cannot touch the grid: retire at once
                        if (P.t_out) P.t_out[r] = -1.0f;
                        if (P.idx_out) { if (P.idx32) ((uint32_t*)P.idx_out)[r] = 0xFFFFFFFFu; else ((unsigned long long*)P.idx_out)[r] = ~0ull; }
                        if (P.shadowed_out) P.shadowed_out[r] = 0;
                        r = ~0ull;
                    }
                }
            }
        }
        if (!__ballot(busy)) {
            if (drained) break;
            continue;
        }
        // ---- walk: every busy lane advances by one slab per step, whatever its level
        bool finished = false;
        for (int s = 0; s < kStepsPerRound; ++s) {
            const bool go = busy && !finished;
            if (!__ballot(go)) break;
            if (go && !walk_step<LDS_MIPS>(R, P, mips_lds)) finished = true;
        }
        // ---- retire: t and the voxel index of the hit; the primitive rank (two dependent loads), the normal and the hit
        // compaction are done by k_rank over all rays afterwards, off this kernel's critical path
        if (finished) {
            const bool hit = R.best_idx != ~0ull;
            if (P.t_out) P.t_out[r] = hit ? R.best : -1.0f;
            if (P.idx_out) { if (P.idx32) ((uint32_t*)P.idx_out)[r] = hit ? (uint32_t)R.best_idx : 0xFFFFFFFFu; else ((unsigned long long*)P.idx_out)[r] = R.best_idx; }
            if (P.shadowed_out) P.shadowed_out[r] = hit ? 1 : 0;
            busy = false;
            r = ~0ull;
        }
    }
}

void launch_walk(const GridParams& g, const TraceMips& mips, const unsigned long long* bricks3, const TraceIO& io, unsigned long long* counter,
                 void* idx_out, bool idx32, hipStream_t s)
{
    const uint64_t nrays = io.nrays;
    if (!nrays) return;
    (void)hipMemsetAsync(counter, 0, sizeof(unsigned long long), s);
    const uint64_t n1 = (uint64_t)mips.d1[0] * mips.d1[1] * mips.d1[2], n2 = (uint64_t)mips.d2[0] * mips.d2[1] * mips.d2[2];
    const uint32_t m1_words = (uint32_t)((n1 + 31) / 32), m2_words = (uint32_t)((n2 + 31) / 32);
    // VOXHIP_TRACE_LDS=0 forces the global-memory mips (the path every grid above ~550^3 takes) -- used by the parity tests
    const char* env_lds = getenv("VOXHIP_TRACE_LDS");
    const bool lds = (size_t)(m1_words + m2_words) * 4 <= 40960 && !(env_lds && atoi(env_lds) == 0);
    const int env_blocks = getenv("VOXHIP_TRACE_BLOCKS") ? atoi(getenv("VOXHIP_TRACE_BLOCKS")) : 0;
    const uint64_t max_blocks = env_blocks > 0 ? (uint64_t)env_blocks : 1024ull * 256ull / VX_W_BLOCK;
    uint64_t nblk = (nrays + VX_W_BLOCK - 1) / VX_W_BLOCK;
    if (nblk > max_blocks) nblk = max_blocks;
    WalkParams P;
    std::memset(&P, 0, sizeof(P));
    P.g = g;
    P.bricks3 = bricks3;
    P.ori_stride = n1 * 8ull;
    P.w1 = mips.w1;
    P.w2 = mips.w2;
    for (int a = 0; a < 3; ++a) { P.d1[a] = mips.d1[a]; P.d2[a] = mips.d2[a]; }
    P.inv_vs = 1.0f / g.vs;
    P.tmin = io.tmin;
    P.any_hit = io.any_hit ? 1 : 0;
    P.m1_words = m1_words;
    P.m2_words = m2_words;
    P.rays = io.rays;
    P.cam = io.cam_dev;
    P.tmax_per_ray = io.tmax_per_ray;
    P.tmax = io.tmax;
    P.idx32 = idx32 ? 1u : 0u;
    P.nrays = nrays;
    P.t_out = io.t_out;
    P.idx_out = idx_out;
    P.shadowed_out = io.shadowed_out;
    P.next_item = counter;
    const size_t shmem = lds ? (size_t)(m1_words + m2_words) * 4 : 0;
    const dim3 grid((unsigned)nblk), block(VX_W_BLOCK);
    if (lds) { VX_KL(k_walk<true>, grid, block, shmem, s, P); } else { VX_KL(k_walk<false>, grid, block, shmem, s, P); }
}

}  // namespace vx
