// vx_trace.hip -- the ray stage around the walk kernel (vx_walk.hip): the occupancy mips of the traversal structure, the per-ray
// post-pass (primitive id, cube-face normal, hit compaction) and the launch sequence.  Replaces what the reference gets from the
// driver around its procedural-hit shader: the acceleration-structure build (hello_vulkan.cpp:737-760), gl_PrimitiveID, and the
// closest-hit stage's normal (raytrace2.rchit:60-73).
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

// Level-1 mip: one bit per brick, x-fastest, from the z-oriented brick words.
__global__ __launch_bounds__(256) void k_brick_mip1(const unsigned long long* __restrict__ bricks, uint64_t nbricks, uint32_t* __restrict__ m1)
{
    // one pass over whole 64-brick groups: the mip -- 64 bits per wave -- is written as two plain words per wave
    const uint64_t ngroups = (nbricks + 63) / 64;
    for (uint64_t gidx = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6; gidx < ngroups; gidx += ((uint64_t)gridDim.x * 256u) >> 6) {
        const uint64_t b = gidx * 64 + (threadIdx.x & 63);
        unsigned long long any = 0;
        if (b < nbricks) {
            const ulonglong2* p = reinterpret_cast<const ulonglong2*>(bricks + b * 8ull);
            const ulonglong2 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            any = q0.x | q0.y | q1.x | q1.y | q2.x | q2.y | q3.x | q3.y;
        }
        const unsigned long long occ = __ballot(any != 0);
        if ((threadIdx.x & 63) == 0) {
            m1[gidx * 2] = (uint32_t)occ;
            m1[gidx * 2 + 1] = (uint32_t)(occ >> 32);
        }
    }
}

void launch_brick_mip1(const unsigned long long* bricks, uint64_t nbricks, uint32_t* m1, hipStream_t s)
{
    if (!nbricks) return;
    uint64_t nblk = (nbricks + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    VX_KL(k_brick_mip1, dim3((unsigned)nblk), dim3(256), 0, s, bricks, nbricks, m1);
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

}  // namespace

// Per-ray post-pass over all rays: primitive id (== gl_PrimitiveID: rank of the voxel in the ascending AABB list), the
// cube-face normal of raytrace2.rchit:60-73, and wavefront hit compaction.
__global__ __launch_bounds__(1024) void k_rank(const float* __restrict__ t, const void* __restrict__ idx_any, int idx32, uint64_t nrays, GridParams g,
                                              const uint32_t* __restrict__ words, const uint32_t* __restrict__ word_prefix, const float* __restrict__ rays,
                                              const Camera* __restrict__ cam, uint32_t* __restrict__ prim_out, float* __restrict__ normal_out, vx_hit* __restrict__ hits,
                                              unsigned long long* nhits, const uint32_t* __restrict__ prefix16 /*optional: word_prefix[16 i], dense*/)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = r < nrays;
    float tt = -1.0f;
    uint32_t prim = 0xFFFFFFFFu;
    if (active) {
        tt = t[r];
        unsigned long long i;  // k_walk writes 32-bit voxel indices when the grid has fewer than 2^32 - 1 voxels, 64-bit ones otherwise
        if (idx32) { const uint32_t i32 = reinterpret_cast<const uint32_t*>(idx_any)[r]; i = i32 == 0xFFFFFFFFu ? ~0ull : (unsigned long long)i32; }
        else i = reinterpret_cast<const unsigned long long*>(idx_any)[r];
        float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
        if (i != ~0ull) {
            const uint64_t wi = i >> 5;
            const uint32_t bit = (uint32_t)i & 31u;
            if (prefix16) {
                // The rank from the voxel's own 64-byte line of the mask and the dense array of every 16th prefix (1/16 of word_prefix: it stays
                // in L2) -- one random line per ray instead of two (word_prefix[wi] is a line of its own): 19.7 -> ... us per 1M rays.
                const uint64_t g0 = wi & ~15ull;
                const uint32_t k = (uint32_t)wi & 15u;
                const uint4* lp = reinterpret_cast<const uint4*>(words + g0);
                const uint4 a = lp[0], b = lp[1], c = lp[2], d = lp[3];
                const uint32_t w16[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
                uint32_t cnt = prefix16[wi >> 4];
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j) cnt += __popc(j < k ? w16[j] : (j == k ? (w16[j] & ((1u << bit) - 1u)) : 0u));
                prim = cnt;
            } else {
                prim = word_prefix[wi] + __popc(words[wi] & ((1u << bit) - 1u));
            }
            if (normal_out) {
                const uint64_t XY = (uint64_t)g.dim[0] * g.dim[1];
                const uint32_t z = (uint32_t)(i / XY);
                const uint64_t rem = i - (uint64_t)z * XY;  // (X * Y may exceed 32 bits on a long thin grid)
                const uint32_t y = (rem >> 32) ? (uint32_t)(rem / g.dim[0]) : (uint32_t)rem / g.dim[0];
                const uint32_t x = (uint32_t)(rem - (uint64_t)y * g.dim[0]);
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

void launch_walk(const GridParams& g, const TraceMips& mips, const TraceIO& io, unsigned long long* counters, int* phase, void* idx_out, bool idx32, hipStream_t s, WalkQueue* queue);

// A gate in front of work that is to run BESIDE a ray batch on another stream (the Vec list's emission, VX_VOXELIZE_LIST_ASYNC): one wave
// that leaves once the ray kernel's work counter has reached `at_least` -- the kernel's persistent workgroups are on the machine (or, with
// the queue's dry value, hand out their last rays) -- or after `timeout_us`, whichever comes first.  The bound makes it safe under tools
// that serialise the kernels of all streams (counter-collecting profilers): a stream-level wait on the counter (hipStreamWaitValue64,
// which polls without a wave) hung three of five rocprofv3 --pmc passes there, because the waiting packet went first and the ray kernel
// was never let through.  The counter is only ever changed by memory-side atomics: read it the same way.
__global__ __launch_bounds__(64) void k_queue_gate(unsigned long long* counter, unsigned long long at_least, unsigned long long timeout_ticks /*100 MHz*/)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = (unsigned long long)wall_clock64();
    for (;;) {
        if (atomicAdd(counter, 0ull) >= at_least) break;
        if ((unsigned long long)wall_clock64() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(32);  // ~0.9 us at 2.4 GHz
    }
}
void launch_queue_gate(const WalkQueue& q, unsigned long long at_least, unsigned timeout_us, hipStream_t s)
{
    if (!q.counter || !at_least) return;
    VX_KL(k_queue_gate, dim3(1), dim3(64), 0, s, q.counter, at_least, (unsigned long long)timeout_us * 100ull);
}

void launch_trace(const GridParams& g, const TraceMips& mips, const uint32_t* word_prefix, const TraceIO& io, unsigned long long* counters /*2*/,
                  int* phase, void* idx_tmp, hipStream_t s, const uint32_t* prefix16, WalkQueue* queue)
{
    const uint64_t nrays = io.nrays;
    if (!nrays) return;
    const bool want_rank = (io.prim_out || io.hits || io.normal_out) && word_prefix && idx_tmp && io.t_out;
    const bool idx32 = trace_idx32(g);
    launch_walk(g, mips, io, counters, phase, want_rank ? idx_tmp : nullptr, idx32, s, queue);
    if (want_rank) {
        if (io.hits && io.nhits) (void)hipMemsetAsync(io.nhits, 0, sizeof(unsigned long long), s);
        const unsigned rthreads = io.hits ? 1024u : 256u;  // the hit list's compaction touches the global counter once per workgroup
        const dim3 rgrid((unsigned)((nrays + rthreads - 1) / rthreads)), rblock(rthreads);
        VX_KL(k_rank, rgrid, rblock, 0, s, io.t_out, (const void*)idx_tmp, idx32 ? 1 : 0, nrays, g, mips.w0, word_prefix, io.rays, io.cam_dev, io.prim_out, io.normal_out,
              io.hits, io.nhits, prefix16);
    }
}

}  // namespace vx
