// vx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the voxelizer / ray hot path.
//
//   K1  k_bbox            vertex min/max with the reference's first-occurrence tie rule   (VoxelBuilder.hpp:198-224)
//   K2a k_tri_setup       de-index triangles, candidate voxel box, work-unit count        (VoxelBuilder.hpp:356-362,170-184)
//   K2  k_voxelize        13-axis SAT per (triangle, y, z, 32-voxel x segment) work unit,
//                         hit bits OR-ed into the occupancy bitmask                       (VoxelBuilder.hpp:118-162/226-335,186-195)
//   K3  k_emit_units      ordered AABB / Morton emission with duplicates (Vec, Octree)    (voxelgridVecEncoding.cpp:19-39, octTree.hpp:765)
//   K4  k_emit_bool       bitmask -> ascending AABB list                                  (voxelgridBool.cpp:18-52)
//   K6  k_trace           two-level conservative 3D-DDA + the rint slab formula           (shaders/raytrace.rint:46-71)
//   scan kernels          device-wide exclusive scan (work-unit bases, popcount prefixes)
//
// Everything here is integer / float32 VALU and HBM/L2 traffic: no MFMA (nothing is a contraction).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (bit-exact float semantics, see vx_math.h).
#include "vx_internal.h"

#include <cstdlib>

#pragma clang fp contract(off)

namespace vx {

static inline unsigned grid_for(uint64_t n, unsigned block, unsigned cap)
{
    uint64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}
// every kernel launch goes through here so that the optional event profiler (vx_prof.cpp) sees it
#define VX_KL(kern, grid, block, shmem, stream, ...)                         \
    do {                                                                     \
        ProfScope ps_(#kern, stream);                                        \
        hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);   \
    } while (0)

constexpr unsigned kMaxBlocks = 256 * 8;  // 256 CUs x 8 resident 256-thread blocks: grid-stride beyond that

// ------------------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_u64_k(unsigned long long v, int src)
{
    return ((unsigned long long)__shfl((unsigned)(v >> 32), src, 64) << 32) | __shfl((unsigned)v, src, 64);
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// K1  bbox.  std::min(cur, x) only replaces cur when x < cur, so the reference keeps the FIRST vertex (in index
// order) among equal-comparing extremes; that only matters for the sign of a zero, and is reproduced by reducing
// 64-bit keys (order-preserving float bits with -0 folded onto +0, then the vertex index) and reading the winning
// vertex's own float back.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned ord_bits(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u << 1) == 0) u = 0;  // -0 -> +0: equal in the reference's comparisons
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// state[0..5]: keys (min x,y,z / max x,y,z), state[6]: number of finished workgroups.  The state is self-cleaning: the last
// workgroup to finish reads the result, writes the six floats and restores the initial values, so a rebuild needs neither a
// host-to-device initialisation nor a second kernel.  `zero64` (optional) is cleared by the same workgroup: the per-build
// setVoxel call counter.
__global__ __launch_bounds__(256) void k_bbox(const float* __restrict__ verts, uint64_t nverts, unsigned long long* state, float* out6,
                                              unsigned long long* zero64, float vs, DevGrid* dgrid)
{
    unsigned long long mn[3] = {~0ull, ~0ull, ~0ull}, mx[3] = {0ull, 0ull, 0ull};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nverts; i += (uint64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const unsigned o = ord_bits(verts[3 * i + a]);
            const unsigned long long kmin = ((unsigned long long)o << 32) | (unsigned)i;
            const unsigned long long kmax = ((unsigned long long)o << 32) | (0xFFFFFFFFu - (unsigned)i);
            mn[a] = kmin < mn[a] ? kmin : mn[a];
            mx[a] = kmax > mx[a] ? kmax : mx[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long o1 = shfl_xor_u64(mn[a], m), o2 = shfl_xor_u64(mx[a], m);
            mn[a] = o1 < mn[a] ? o1 : mn[a];
            mx[a] = o2 > mx[a] ? o2 : mx[a];
        }
    }
    // one set of atomics per workgroup (every wave hitting the same six addresses serialises: 147 us -> a few us)
    __shared__ unsigned long long red[4][6];
    __shared__ int last_s;
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[wv][a] = mn[a]; red[wv][3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        unsigned long long v = red[0][a];
        for (int w = 1; w < 4; ++w) {
            const unsigned long long o = red[w][a];
            v = a < 3 ? (o < v ? o : v) : (o > v ? o : v);
        }
        if (a < 3) atomicMin(&state[a], v); else atomicMax(&state[a], v);
    }
    // The six min/max atomics and the ticket below all execute at the memory side: the ticket must only not overtake them, i.e. be
    // issued after they were acknowledged (s_waitcnt vmcnt(0) by the wave that sent them, in front of the barrier).  No cache has
    // anything to write back or to invalidate for them: the two agent-scope fences that stood here cost 4 us of this 15 us kernel.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last_s = atomicAdd(&state[6], 1ull) == (unsigned long long)gridDim.x - 1ull;
    __syncthreads();
    if (!last_s) return;
    float val = 0.0f;
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        const unsigned long long key = __hip_atomic_load(&state[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nverts == 0) {
            val = a < 3 ? INFINITY : -INFINITY;
        } else {
            const unsigned low = (unsigned)key;
            const unsigned idx = a < 3 ? low : 0xFFFFFFFFu - low;
            val = verts[3 * (uint64_t)idx + (a % 3)];
        }
        out6[a] = val;
        __hip_atomic_store(&state[a], a < 3 ? ~0ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 6) __hip_atomic_store(&state[6], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (zero64 && threadIdx.x >= 64u && threadIdx.x < 64u + kCallCounters) zero64[(threadIdx.x - 64u) * 8u] = 0ull;
    if (dgrid && threadIdx.x < 64) {
        // the grid the host will derive from this bbox (VoxelBuilder.hpp:347-349: origin = bbox min, dim = ceil((max - min) / vs),
        // same IEEE division), for the kernels that are queued before the host has seen the bbox
        const float mx = __shfl(val, (int)(threadIdx.x + 3u) & 63, 64);
        if (threadIdx.x < 3) {
            uint32_t d = 0;
            if (nverts) {
                const float q = ceilf((mx - val) / vs);
                d = (q >= 0.0f && q <= (float)kMaxDim) ? (uint32_t)q : 0u;  // out of range: the host reports the error, the device sees an empty grid
            }
            dgrid->org[threadIdx.x] = val;
            dgrid->dim[threadIdx.x] = d;
        }
    }
}

#ifndef VX_BBOX_PER_THREAD
#define VX_BBOX_PER_THREAD 8
#endif
void bbox_state_init(unsigned long long state7[7])
{
    for (int a = 0; a < 3; ++a) { state7[a] = ~0ull; state7[3 + a] = 0ull; }
    state7[6] = 0ull;
}

void launch_bbox(const float* verts, uint64_t nverts, unsigned long long* state7, float* out6, unsigned long long* zero64, hipStream_t s, float vs,
                 DevGrid* dgrid)
{
    VX_KL(k_bbox, dim3(nverts ? grid_for(nverts, 256 * VX_BBOX_PER_THREAD, 512) : 1u), dim3(256), 0, s, verts, nverts, state7, out6, zero64, vs, dgrid);
}

// ------------------------------------------------------------------------------------------------------------
// Device-wide exclusive scan of uint32 (optionally of their popcounts): out[i] = sum_{j<i} f(in[j]) for i in [0,n].
// Three passes: tile sums -> spine (one block) -> apply.  Sums are carried in 64 bits so the caller can detect a
// total that does not fit the 32-bit outputs.
// ------------------------------------------------------------------------------------------------------------
template <bool POPC>
__device__ __forceinline__ unsigned scan_ld(const uint32_t* __restrict__ in, uint64_t i, uint64_t n)
{
    if (i >= n) return 0u;
    const unsigned v = in[i];
    return POPC ? (unsigned)__popc(v) : v;
}

__device__ __forceinline__ unsigned block_excl_scan_256(unsigned v, unsigned* lds /*>=4*/, unsigned& block_total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds[wv] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned s = lds[w];
        if (w < wv) base += s;
        tot += s;
    }
    block_total = tot;
    __syncthreads();
    return base + inc - v;
}

template <bool POPC>
__global__ __launch_bounds__(kScanBlock) void k_scan_sums(const uint32_t* __restrict__ in, uint64_t n, unsigned long long* sums)
{
    __shared__ unsigned long long lds[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) s += scan_ld<POPC>(in, base + j, n);
    s = wave_sum_u64(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(1024) void k_scan_spine(unsigned long long* sums, uint32_t nblocks, unsigned long long* total)
{
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < nblocks ? sums[i] : 0ull;
        unsigned long long inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned lo = (unsigned)inc, hi = (unsigned)(inc >> 32);
            lo = __shfl_up(lo, d, 64);
            hi = __shfl_up(hi, d, 64);
            if (lane >= d) inc += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        unsigned long long wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wv) wbase += wsum[w];
            tot += wsum[w];
        }
        const unsigned long long carry = carry_s;
        if (i < nblocks) sums[i] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}

template <bool POPC>
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                                                          const unsigned long long* __restrict__ sums)
{
    __shared__ unsigned lds[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    unsigned v[kScanItems];
    unsigned tsum = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) { v[j] = scan_ld<POPC>(in, base + j, n); tsum += v[j]; }
    unsigned btot;
    unsigned pre = block_excl_scan_256(tsum, lds, btot) + (unsigned)sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (base + j <= n) out[base + j] = pre;
        pre += v[j];
    }
}

// Single-pass variant (decoupled look-back): one kernel instead of three.  Tiles are handed out by a ticket counter so that
// every tile a block waits for has already started; a tile publishes (flag | value) in ONE 64-bit word -- flag 1: the tile's
// own sum ("aggregate"), flag 2: the sum of everything up to and including the tile ("prefix") -- and wave 0 of the block
// walks back over its predecessors' words, 64 at a time, until it meets a prefix.  Values are carried in 62 bits.
// Flag and value travel in the same word, so relaxed agent-scope atomics are enough (acquire/release would add a cache
// invalidate / write-back to every poll: measured 190 us per scan instead of 40).
constexpr unsigned long long kScanFlagA = 1ull << 62, kScanFlagP = 2ull << 62, kScanValMask = (1ull << 62) - 1ull;
// GENERATION MODE (gen != 0; scans of at most kScanGenTiles tiles, i.e. every workgroup of the launch is resident at once):
//   * tiles are taken in blockIdx order instead of by ticket -- a workgroup never waits for one that is not running yet, and the
//     returning atomic in front of everything else (~2 us) is gone;
//   * a state word carries the scan's generation number in bits 40..61 (value in bits 0..39): a word of an older scan on the same
//     buffer reads as "not there yet", so nothing has to be cleaned up -- the finished-tiles counter, its returning atomic and the
//     zeroing loop are gone as well.  The caller numbers the scans of a buffer 1, 2, 3, ... (and clears the buffer when the number
//     wraps at 2^22); a buffer fresh from the allocator is cleared once.
#ifndef VX_SCAN_GEN_TILES
#define VX_SCAN_GEN_TILES 512
#endif
constexpr uint32_t kScanGenTiles = VX_SCAN_GEN_TILES;
constexpr unsigned long long kScanGenValMask = (1ull << 40) - 1ull;

// Large tiles (1024 threads x 16 elements): a 4M-element scan is 256 tiles, so the look-back chain is a handful of hops
// (with 2048-element tiles the chain of 2048 hops at cross-XCD atomic latency cost as much as the three-pass scan).
#ifndef VX_SCAN_BLOCK
#define VX_SCAN_BLOCK 1024
#endif
#ifndef VX_SCAN_ITEMS
#define VX_SCAN_ITEMS 16
#endif
constexpr int kOneBlock = VX_SCAN_BLOCK, kOneItems = VX_SCAN_ITEMS, kOneTile = kOneBlock * kOneItems;

// MODE 0: the uint32 values themselves, 1: their popcounts, 2: `in` is an array of BYTES (n of them), sixteen per 16-byte load
template <int MODE>
__global__ __launch_bounds__(kOneBlock) void k_scan_onepass(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n,
                                                           unsigned long long* status /*[0]: ticket, [1]: finished tiles, [2 + tile]: state*/, uint32_t ntiles,
                                                           unsigned long long* total, unsigned long long total_tag /*OR-ed into *total: bits 48..63*/,
                                                           uint32_t* __restrict__ sel /*optional: sel[c] = index of the element whose range [pre, pre + v) holds c * 1024*/,
                                                           uint32_t gen /*0: tickets + self-cleaning state; else generation mode*/,
                                                           uint32_t* __restrict__ group16 /*optional: group16[i] = out[16 i], a dense copy of every 16th output*/)
{
    __shared__ unsigned wsum[kOneBlock / 64];
    __shared__ unsigned tile_s;
    __shared__ unsigned long long prefix_s;
    unsigned tile = blockIdx.x;
    if (!gen) {
        if (threadIdx.x == 0) tile_s = (unsigned)atomicAdd(status, 1ull);
        __syncthreads();
        tile = tile_s;
    }
    const unsigned long long gtag = (unsigned long long)gen << 40;                       // this scan's words: flag | gtag | value
    const unsigned long long vmask = gen ? kScanGenValMask : kScanValMask;
    const unsigned long long fmask = gen ? (3ull << 62) | (((1ull << 22) - 1ull) << 40) : (3ull << 62);  // what must match for a word to count
    unsigned long long* st = status + 2;
    const uint64_t base = (uint64_t)tile * kOneTile + (uint64_t)threadIdx.x * kOneItems;
    unsigned v[kOneItems];
    const bool full = base + kOneItems <= n;
    constexpr bool POPC = MODE == 1;
    if (MODE == 2) {
        static_assert(kOneItems == 16, "one 16-byte load per thread");
        const uint8_t* inb = reinterpret_cast<const uint8_t*>(in);
        if (full) {
            const uint4 a = *reinterpret_cast<const uint4*>(inb + base);
            const uint32_t w4[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < kOneItems; ++j) v[j] = (w4[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        } else {
#pragma unroll
            for (int j = 0; j < kOneItems; ++j) v[j] = base + j < n ? (unsigned)inb[base + j] : 0u;
        }
    } else if (full) {
#pragma unroll
        for (int q = 0; q < kOneItems / 4; ++q) {
            const uint4 a = *reinterpret_cast<const uint4*>(in + base + 4 * q);
            v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
        }
        if (POPC) {
#pragma unroll
            for (int j = 0; j < kOneItems; ++j) v[j] = (unsigned)__popc(v[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < kOneItems; ++j) v[j] = scan_ld<POPC>(in, base + j, n);
    }
    unsigned tsum = 0;
#pragma unroll
    for (int j = 0; j < kOneItems; ++j) tsum += v[j];
    // exclusive scan of the thread sums over the block
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned inc = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned wbase = 0, btot = 0;
#pragma unroll
    for (int w = 0; w < kOneBlock / 64; ++w) {
        const unsigned x = wsum[w];
        if (w < wv) wbase += x;
        btot += x;
    }
    unsigned pre = wbase + inc - tsum;
    if (threadIdx.x < 64) {
        unsigned long long excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(&st[0], kScanFlagP | gtag | (unsigned long long)btot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(&st[tile], kScanFlagA | gtag | (unsigned long long)btot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long look = (long long)tile - 1;
            for (;;) {
                const long long idx = look - lane;
                unsigned long long w;
                bool there;
                do {  // every predecessor in the window has at least started (ticket order / all resident): wait for its first word
                    w = idx >= 0 ? __hip_atomic_load(&st[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (kScanFlagP | gtag);
                    // (a word of this scan: flag A or P and, in generation mode, this scan's number)
                    there = ((w & fmask) == (kScanFlagA | gtag)) || ((w & fmask) == (kScanFlagP | gtag));
                } while (__ballot(!there));
                const unsigned long long pm = __ballot((w >> 62) == 2ull);
                const int first_p = pm ? __ffsll((long long)pm) - 1 : 64;
                excl += wave_sum_u64(lane <= first_p ? (w & vmask) : 0ull);
                if (pm) break;
                look -= 64;
            }
            // (generation mode keeps 40 value bits: sums saturate there instead of running into the generation number -- a total that
            // large is far beyond what the 32-bit outputs can address and is reported as a capacity error by every caller)
            if (excl + (unsigned long long)btot > vmask) excl = vmask - (unsigned long long)btot;
            if (lane == 0) __hip_atomic_store(&st[tile], kScanFlagP | gtag | (excl + (unsigned long long)btot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            prefix_s = excl;
            if (tile == ntiles - 1 && total) *total = total_tag | (excl + (unsigned long long)btot);
        }
    }
    __syncthreads();
    pre += (unsigned)prefix_s;
    static_assert(kOneItems == 16, "group16: one group per thread");
    if (group16 && base <= n) group16[base >> 4] = pre;
    if (full) {  // all outputs of this thread exist (out has n + 1 entries)
#pragma unroll
        for (int q = 0; q < kOneItems / 4; ++q) {
            uint4 a;
            a.x = pre; pre += v[4 * q]; a.y = pre; pre += v[4 * q + 1]; a.z = pre; pre += v[4 * q + 2]; a.w = pre; pre += v[4 * q + 3];
            *reinterpret_cast<uint4*>(out + base + 4 * q) = a;
            if (sel) {  // (values of at most 1024: an element's range holds at most one multiple of 1024)
                const uint32_t p4[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t ve = v[4 * q + e], c = (p4[e] + ve - 1u) >> 10;
                    if (ve && (c << 10) >= p4[e]) sel[c] = (uint32_t)(base + 4 * q + e);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < kOneItems; ++j) {
            if (base + j <= n) out[base + j] = pre;
            if (sel && v[j] && base + j < n) {
                const uint32_t c = (pre + v[j] - 1u) >> 10;
                if ((c << 10) >= pre) sel[c] = (uint32_t)(base + j);
            }
            pre += v[j];
        }
    }
    // self-cleaning: the tile that finishes its look-back last (nobody reads a state word any more) zeroes the ticket, the
    // counter and every state word, so the next scan on this buffer needs no memset.  Last thing the workgroup does: the
    // counter's old value takes microseconds to come back and nothing else has to wait for it.
    if (!gen && threadIdx.x < 64) {
        unsigned long long fin = 0;
        if (lane == 0) fin = atomicAdd(&status[1], 1ull);
        fin = shfl_u64_k(fin, 0);
        if (fin == (unsigned long long)ntiles - 1ull) {
            for (uint32_t i = lane; i < ntiles + 2u; i += 64u) __hip_atomic_store(&status[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

size_t scan_tmp_bytes(uint64_t n) { return (size_t)(((n + 1) + kScanTile - 1) / kScanTile + 4) * sizeof(unsigned long long); }

bool launch_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, bool popcount_input, void* tmp, unsigned long long* total64,
                     hipStream_t s, bool tmp_is_zero, unsigned long long total_tag, uint32_t* sel1024, uint32_t gen, uint32_t* group16)
{
    const uint32_t nblocks = (uint32_t)(((n + 1) + kScanTile - 1) / kScanTile);
    unsigned long long* status = (unsigned long long*)tmp;
    static const bool three_pass = getenv("VOXHIP_SCAN_3PASS") && atoi(getenv("VOXHIP_SCAN_3PASS"));
    const bool aligned = ((((uintptr_t)in) | ((uintptr_t)out)) & 15u) == 0;  // the single-pass kernel moves 16-byte vectors
    if (!three_pass && aligned) {
        const uint32_t ntiles = (uint32_t)(((n + 1) + kOneTile - 1) / kOneTile);
        // generation mode: small scans only (every tile resident).  A larger scan on a generation-managed buffer clears its state words
        // first (they hold older scans' words) and runs with tickets, which leaves them zero again.
        const uint32_t g = (gen && ntiles <= kScanGenTiles) ? gen : 0u;
        if (!g && (!tmp_is_zero || gen)) (void)hipMemsetAsync(status, 0, (size_t)(ntiles + 2) * sizeof(unsigned long long), s);
        if (popcount_input) VX_KL(k_scan_onepass<1>, dim3(ntiles), dim3(kOneBlock), 0, s, in, out, n, status, ntiles, total64, total_tag, sel1024, g, group16);
        else VX_KL(k_scan_onepass<0>, dim3(ntiles), dim3(kOneBlock), 0, s, in, out, n, status, ntiles, total64, total_tag, sel1024, g, group16);
        return true;
    }
    unsigned long long* sums = status;
    if (popcount_input) {
        VX_KL(k_scan_sums<true>, dim3(nblocks), dim3(kScanBlock), 0, s, in, n, sums);
        VX_KL(k_scan_spine, dim3(1), dim3(1024), 0, s, sums, nblocks, total64);
        VX_KL(k_scan_apply<true>, dim3(nblocks), dim3(kScanBlock), 0, s, in, out, n, sums);
    } else {
        VX_KL(k_scan_sums<false>, dim3(nblocks), dim3(kScanBlock), 0, s, in, n, sums);
        VX_KL(k_scan_spine, dim3(1), dim3(1024), 0, s, sums, nblocks, total64);
        VX_KL(k_scan_apply<false>, dim3(nblocks), dim3(kScanBlock), 0, s, in, out, n, sums);
    }
    if (tmp_is_zero || gen) (void)hipMemsetAsync(status, 0, scan_tmp_bytes(n), s);  // keep the caller's "zero between scans" contract (no stale words either)
    return false;  // (*total64 carries no tag)
}

// exclusive scan of n BYTES into out[0..n] (uint32); single-pass kernel only (in and out 16-byte aligned, tmp all zero before and after)
void launch_scan_u8(const uint8_t* in, uint32_t* out, uint64_t n, void* tmp, unsigned long long* total64, hipStream_t s, unsigned long long total_tag, uint32_t gen)
{
    const uint32_t ntiles = (uint32_t)(((n + 1) + kOneTile - 1) / kOneTile);
    const uint32_t g = (gen && ntiles <= kScanGenTiles) ? gen : 0u;
    if (!g && gen) (void)hipMemsetAsync(tmp, 0, (size_t)(ntiles + 2) * sizeof(unsigned long long), s);
    VX_KL(k_scan_onepass<2>, dim3(ntiles), dim3(kOneBlock), 0, s, reinterpret_cast<const uint32_t*>(in), out, n, (unsigned long long*)tmp, ntiles, total64, total_tag,
          (uint32_t*)nullptr, g, (uint32_t*)nullptr);
}

// ------------------------------------------------------------------------------------------------------------
// K2a  triangle setup.  One lane per triangle: gather the three vertices, compute the candidate voxel box exactly
// as the reference does, clamp it to this rank's z slab, and count work units.  A work unit is one row of the box
// (fixed y, z) cut at multiples of 32 in x, i.e. at most one 32-voxel stretch that maps onto <= 2 bitmask words.
// ------------------------------------------------------------------------------------------------------------
// Exact trimming of a candidate range.  The reference's range (VoxelBuilder.hpp:175-184) carries a spare cell on the high
// side of every axis (`+ 2`), i.e. slabs of voxels that its SAT then rejects by the box-axis test of that axis (:94-100).
// That test depends only on the slab's index, so a slab for which it separates -- evaluated here with the very floats the
// SAT uses (cell_centre, v - c, min3/max3 against half) -- cannot contain a hit and is dropped from the end of the range.
// On the bench mesh this halves the candidate volume (4.2 x 4.7 x 4.7 -> 3.2 x 3.7 x 3.7 voxels per triangle).
__device__ __forceinline__ void trim_axis(float a0, float a1, float a2, float org, float vs, float half, int& s, int& e)
{
    while (e > s) {
        const float c = cell_centre(org, vs, (uint32_t)(e - 1));
        const float p0 = a0 - c, p1 = a1 - c, p2 = a2 - c;
        if (!((min3(p0, p1, p2) > half) || (max3(p0, p1, p2) < -half))) break;
        --e;
    }
    while (e > s) {
        const float c = cell_centre(org, vs, (uint32_t)s);
        const float p0 = a0 - c, p1 = a1 - c, p2 = a2 - c;
        if (!((min3(p0, p1, p2) > half) || (max3(p0, p1, p2) < -half))) break;
        ++s;
    }
}

__global__ __launch_bounds__(256) void k_tri_setup(const float* __restrict__ verts, const int32_t* __restrict__ idx, uint64_t tri_begin,
                                                   uint32_t ntri, GridParams g, float vsize, uint32_t zlo, uint32_t zhi,
                                                   TriRec* __restrict__ recs, uint32_t* __restrict__ units, const DevGrid* __restrict__ dgrid,
                                                   uint4* __restrict__ clear /*optional: 16-byte pieces to zero*/, uint64_t clear_n,
                                                   uint64_t shard_wb, uint64_t shard_we /*with dgrid: the bitmask words of a sharded build (0, 0: all)*/,
                                                   uint32_t* __restrict__ ext /*bits 16..20 of the six range values (grids with an axis above 65535 cells)*/,
                                                   uint32_t shard_rank, uint32_t shard_world /*with dgrid, world > 1: the word shard of that rank*/)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    // (optional) the build's bitmask is cleared by this kernel's threads, beside their own work: a launch and its gap less
    for (uint64_t i = t; i < clear_n; i += (uint64_t)gridDim.x * 256u) clear[i] = make_uint4(0u, 0u, 0u, 0u);
    if (t >= ntri) return;
    if (dgrid) {  // origin and dims from the bbox kernel queued in front (the host has not seen them yet); whole grid in z
#pragma unroll
        for (int a = 0; a < 3; ++a) { g.org[a] = dgrid->org[a]; g.dim[a] = dgrid->dim[a]; }
        zlo = 0u;
        zhi = g.dim[2];
        if (shard_world > 1u) {  // word shard by rank: vx_shard_words on the device-side dims (the host redoes it once it has seen the bbox)
            const uint64_t nw = ((uint64_t)g.dim[0] * g.dim[1] * g.dim[2] + 31ull) / 32ull;
            const uint64_t chunk = (nw + shard_world - 1u) / shard_world;
            shard_wb = (uint64_t)shard_rank * chunk;
            if (shard_wb > nw) shard_wb = nw;
            shard_we = shard_wb + chunk > nw ? nw : shard_wb + chunk;
            if (shard_we == shard_wb) { zlo = zhi = 0u; shard_we = 0; }  // an empty shard (more ranks than words)
        }
        if (shard_we) {  // the z slab that holds the voxels of words [shard_wb, shard_we): the host's own arithmetic (vx_voxelize_into)
            const uint64_t XY = (uint64_t)g.dim[0] * g.dim[1];
            if (XY) {
                zlo = (uint32_t)((shard_wb * 32ull) / XY);
                const uint64_t zh = (shard_we * 32ull + XY - 1ull) / XY;
                zhi = zh > g.dim[2] ? g.dim[2] : (uint32_t)zh;
                if (zlo > zhi) zlo = zhi;
            }
        }
    }
    const int32_t* ip = idx + 3 * (tri_begin + t);
    TriRec r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint64_t vi = (uint64_t)(uint32_t)ip[k];
        r.v[3 * k + 0] = verts[3 * vi + 0];
        r.v[3 * k + 1] = verts[3 * vi + 1];
        r.v[3 * k + 2] = verts[3 * vi + 2];
    }
    int xs, xe, ys, ye, zs, ze;
    cand_axis(r.v[0], r.v[3], r.v[6], g.org[0], vsize, g.dim[0], xs, xe);
    cand_axis(r.v[1], r.v[4], r.v[7], g.org[1], vsize, g.dim[1], ys, ye);
    cand_axis(r.v[2], r.v[5], r.v[8], g.org[2], vsize, g.dim[2], zs, ze);
    zs = zs > (int)zlo ? zs : (int)zlo;
    ze = ze < (int)zhi ? ze : (int)zhi;
    trim_axis(r.v[0], r.v[3], r.v[6], g.org[0], g.vs, g.half, xs, xe);
    trim_axis(r.v[1], r.v[4], r.v[7], g.org[1], g.vs, g.half, ys, ye);
    trim_axis(r.v[2], r.v[5], r.v[8], g.org[2], g.vs, g.half, zs, ze);
    const uint32_t nx = xe > xs ? (uint32_t)(xe - xs) : 0u;
    const uint32_t ny = ye > ys ? (uint32_t)(ye - ys) : 0u;
    const uint32_t nz = ze > zs ? (uint32_t)(ze - zs) : 0u;
    uint32_t u = 0;
    if (nx && ny && nz) {
        const uint32_t nseg = (((uint32_t)xs + nx - 1u) >> 5) - ((uint32_t)xs >> 5) + 1u;
        const uint64_t uu = (uint64_t)nseg * ny * nz;
        u = uu > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)uu;
    }
    // start | count << 16 per axis, low 16 bits each; bits 16..20 (cells 65536 .. 2^21 - 1) go to the extension word, which the unit
    // kernels only read for grids that have such an axis
    const uint32_t xs_ = nx ? (uint32_t)xs : 0u, ys_ = ny ? (uint32_t)ys : 0u, zs_ = nz ? (uint32_t)zs : 0u;
    r.xr = (xs_ & 0xFFFFu) | (nx << 16);
    r.yr = (ys_ & 0xFFFFu) | (ny << 16);
    r.zr = (zs_ & 0xFFFFu) | (nz << 16);
    if (ext) ext[t] = (xs_ >> 16) | ((nx >> 16) << 5) | ((ys_ >> 16) << 10) | ((ny >> 16) << 15) | ((zs_ >> 16) << 20) | ((nz >> 16) << 25);
    recs[t] = r;
    units[t] = u;
}

void launch_tri_setup(const float* verts, const int32_t* idx, uint64_t tri_begin, uint32_t ntri, const GridParams& g, int sat_variant,
                      uint32_t zlo, uint32_t zhi, TriRec* recs, uint32_t* units, hipStream_t s, const DevGrid* dgrid, void* clear, uint64_t clear_bytes, uint64_t shard_wb, uint64_t shard_we,
                      uint32_t* ext, uint32_t shard_rank, uint32_t shard_world)
{
    if (!ntri) return;
    // serial driver: voxelSize = halfVoxelSize.x * 2.0f (VoxelBuilder.hpp:173); threaded driver: vSize = voxelSize (:500)
    const float vsize = sat_variant == 0 ? g.half * 2.0f : g.vs;
    VX_KL(k_tri_setup, dim3((ntri + 255) / 256), dim3(256), 0, s, verts, idx, tri_begin, ntri, g, vsize, zlo, zhi, recs, units, dgrid,
          reinterpret_cast<uint4*>(clear), clear ? clear_bytes / 16 : 0, shard_wb, shard_we, ext, shard_rank, shard_world);
}

// ------------------------------------------------------------------------------------------------------------
// Work-unit decode shared by K2 and K3: unit u -> triangle t (largest t with unit_base[t] <= u) and the row
// segment it covers.
// ------------------------------------------------------------------------------------------------------------
struct Unit {
    uint32_t tri;
    uint32_t x0, x1;   // voxel x range [x0, x1), inside one 32-aligned stretch
    uint32_t xseg;     // x0 & ~31
    uint32_t y, z;
};

__device__ __forceinline__ uint32_t find_tri(const uint32_t* __restrict__ unit_base, uint32_t ntri, uint32_t u)
{
    uint32_t lo = 0, hi = ntri;  // unit_base[lo] <= u < unit_base[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (unit_base[mid] <= u) lo = mid; else hi = mid;
    }
    return lo;
}

// a / b for the unit decode, a < 2^22, 1 <= b < 2^24.  The compiler's 32-bit division is ~20 instructions, four of them quarter-rate
// multiplies, and a unit needs two; here the float quotient a * rcp(b) is within one of the true one (relative error < 2^-22),
// which one 24-bit multiply-subtract tells and repairs.
__device__ __forceinline__ uint32_t udiv_unit(uint32_t a, uint32_t b)
{
    uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)b));
    const int rem = (int)a - (int)__umul24(q, b);  // q <= a / b + 1: q * b <= a + b fits easily
    if (rem < 0) --q;
    else if (rem >= (int)b) ++q;
    return q;
}

// e: the triangle's extension word (k_tri_setup), 0 for grids without an axis above 65535 cells
__device__ __forceinline__ Unit decode_unit(const TriRec& r, uint32_t tri, uint32_t rel, uint32_t e = 0u)
{
    Unit w;
    w.tri = tri;
    const uint32_t xs = (r.xr & 0xFFFFu) | ((e & 31u) << 16), nx = (r.xr >> 16) | (((e >> 5) & 31u) << 16);
    const uint32_t ys = (r.yr & 0xFFFFu) | (((e >> 10) & 31u) << 16), ny = (r.yr >> 16) | (((e >> 15) & 31u) << 16);
    const uint32_t zs = (r.zr & 0xFFFFu) | (((e >> 20) & 31u) << 16);
    const uint32_t seg0 = xs >> 5;
    const uint32_t nseg = ((xs + nx - 1u) >> 5) - seg0 + 1u;
    uint32_t row, sx, zz, yy;
#ifndef VX_UNIT_INT_DIV
    if (rel < (1u << 22)) {  // every triangle but the wall-sized ones of the largest grids (nseg <= 2^16, ny <= 2^21)
        row = nseg == 1u ? rel : udiv_unit(rel, nseg);
        sx = rel - __umul24(row, nseg);
        zz = udiv_unit(row, ny);
        yy = row - __umul24(zz, ny);
    } else
#endif
    {
        row = rel / nseg; sx = rel - row * nseg;
        zz = row / ny; yy = row - zz * ny;
    }
    w.xseg = (seg0 + sx) << 5;
    w.x0 = xs > w.xseg ? xs : w.xseg;
    const uint32_t xe = xs + nx, se = w.xseg + 32u;
    w.x1 = xe < se ? xe : se;
    w.y = ys + yy;
    w.z = zs + zz;
    return w;
}

__device__ __forceinline__ TriRec load_rec(const TriRec* __restrict__ recs, uint32_t t)
{
    const float4* p = reinterpret_cast<const float4*>(recs + t);
    const float4 a = p[0], b = p[1], c = p[2];
    TriRec r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = c.x;
    r.xr = __float_as_uint(c.y); r.yr = __float_as_uint(c.z); r.zr = __float_as_uint(c.w);
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// Unit -> triangle lookup without a per-lane global binary search (K2 was latency-bound on it: 61 % of wave cycles in
// s_waitcnt).  k_unit_blocks records, for every 64-unit block, the triangle that owns the block's first unit; a WAVE then
// stages the unit bases and the 48-byte records of the (few) triangles its 64 units belong to in its own piece of LDS and
// every lane searches there.  Blocks that span too many triangles (long runs of zero-unit triangles) fall back to the
// global search.  Nothing in the scheme synchronises waves with each other.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kStageTris = 64;
constexpr uint32_t kUnitBlockLog2 = 6;

// One thread per 64-unit block: binary search of the unit bases for the triangle that owns the block's first unit (the
// largest t with unit_base[t] <= 64 b; triangles without units share their base with the next one and are skipped by taking
// the LAST such t).  The previous form -- one thread per triangle writing the blocks it spans -- serialised on the few
// triangles that span hundreds of blocks.
__global__ __launch_bounds__(256) void k_unit_blocks(const uint32_t* __restrict__ unit_base, uint32_t ntri, uint32_t* __restrict__ block_tri,
                                                     uint32_t cap_blocks /*entries block_tri can hold*/)
{
    const uint32_t U = unit_base[ntri];
    const uint32_t nUB = (U + 63u) >> kUnitBlockLog2;
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= nUB || b >= cap_blocks) return;
    const uint32_t u = b << kUnitBlockLog2;
    uint32_t lo = 0, hi = ntri;  // unit_base[lo] <= u < unit_base[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (unit_base[mid] <= u) lo = mid; else hi = mid;
    }
    block_tri[b] = lo;
}

// total_units: the host's figure -- or, for a launch queued before the host knows it, the most the table can describe (the kernel
// reads the real total from unit_base[ntri] and never writes at or beyond cap_blocks)
void launch_unit_blocks(const uint32_t* unit_base, uint32_t ntri, uint32_t total_units, uint32_t* block_tri, hipStream_t s, uint32_t cap_blocks)
{
    const uint32_t nUB = (uint32_t)(((uint64_t)total_units + 63u) >> kUnitBlockLog2);
    if (!ntri || !nUB) return;
    VX_KL(k_unit_blocks, dim3((nUB + 255) / 256), dim3(256), 0, s, unit_base, ntri, block_tri, cap_blocks);
}

#ifdef VX_VOX_DEBUG
// diagnostic build: wave cycles per phase of for_each_unit: [4] wait for outstanding memory operations at the top of a pass,
// [0] issue of the next pass's staging, [1] unit search + record read, [2] the functor (SAT / emission), [3] rest
__device__ unsigned long long g_vox_dbg[8];
#define VX_V_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); vdbg[i] += now_ - vlast; vlast = now_; }
#define VX_V_SENT ++sent;
#else
#define VX_V_T(i)
#define VX_V_SENT
#endif

struct __attribute__((aligned(16))) UnitStage {  // one wave's staging buffer
    float4 rec[kStageTris * 3];
    uint32_t base[kStageTris];
};
constexpr uint32_t kStagesPerBlock = 2 * 4;  // double-buffered, four waves per 256-thread workgroup
// Grid of k_voxelize.  While the kernel was bound by the memory side's request rate, one even, fully resident set of persistent
// workgroups was best (256 CUs x 6, the number the 26 KiB of double staging buffers allowed: 110 -> 104 us against 256 x 8).  Bound by
// VALU issue (round 3, tiled build mask) it wants a finer split of the passes and does not need the second staging buffer: one buffer
// per wave (13 KiB) and 3072 workgroups 63.0 us, against 69.4 (two buffers, 1536), 68.1 (one, 1536), 66.2 (two, 3072), 64.4 (one,
// 4096), 71.4 (one, 6144); forcing eight waves per SIMD (64 VGPRs) with 2048 / 4096 workgroups 63.4 / 64.0.
#ifndef VX_UNIT_BLOCKS
#define VX_UNIT_BLOCKS (256 * 12)
#endif
constexpr unsigned kUnitBlocks = VX_UNIT_BLOCKS;
#ifndef VX_EMIT_BLOCKS
#define VX_EMIT_BLOCKS (2u * kMaxBlocks)
#endif
#ifndef VX_EMIT_NBUF
#define VX_EMIT_NBUF 1
#endif

// Asynchronous global -> LDS copies (gfx950 LDS-DMA: no VGPR destination).  The LDS destination of one wave-instruction is
// lds_base + lane * size, so both images are lane-linear.
typedef const void __attribute__((address_space(1)))* vx_gptr;
typedef void __attribute__((address_space(3)))* vx_lptr;
__device__ __forceinline__ void stage_issue(UnitStage& S, const TriRec* __restrict__ recs, const uint32_t* __restrict__ unit_base, uint32_t t_lo, uint32_t n,
                                            uint32_t lane)
{
    if (lane < n)  // n <= 64 bases (the search never reads base[n]): one 4-byte piece per lane
        __builtin_amdgcn_global_load_lds((vx_gptr)(unit_base + t_lo + lane), (vx_lptr)&S.base[0], 4, 0, 0);
    const float4* src = reinterpret_cast<const float4*>(recs + t_lo);
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {  // 3 n <= 192 sixteen-byte pieces
        const uint32_t i = lane + k * 64u;
        if (i < n * 3u) __builtin_amdgcn_global_load_lds((vx_gptr)(src + i), (vx_lptr)&S.rec[k * 64u], 16, 0, 0);
    }
}

// Calls f(u, triangle, record, u - unit_base[triangle], valid) for every work unit, one unit per lane, 64 per wave pass: the 64
// units of block u >> 6.  ALL lanes of the wave make the call, in uniform control flow (the functor may use wave-wide operations);
// `valid` is false for the lanes beyond the last unit.  Each
// wave works on its own: it stages through its own two LDS buffers and never meets a workgroup barrier, so a wave with a
// long row (32 voxels, all axes) does not hold up its neighbours.  The staging is software-pipelined: a pass first reads
// its lane's record out of the buffer that landed, THEN issues the LDS-DMA of the next pass into the other buffer and the
// scalar loads of the range of the pass after it, and only then runs the functor -- so the one s_waitcnt vmcnt(0) at the top
// of a pass finds the copies complete.  (The order matters: the compiler orders every LDS read after all LDS-DMA in flight,
// whichever buffer it targets, and a range load that feeds arithmetic is waited for where the arithmetic stands.)
template <int NBUF = 2, class F>
__device__ __forceinline__ void for_each_unit(const TriRec* __restrict__ recs, const uint32_t* __restrict__ unit_base,
                                              const uint32_t* __restrict__ block_tri, uint32_t ntri, UnitStage* stages /*[NBUF * 4]*/, F&& f)
{
    const uint32_t U = unit_base[ntri];
    const uint32_t nUB = (U + 63u) >> kUnitBlockLog2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: the range loads go scalar
    const uint32_t stride = gridDim.x * (blockDim.x >> 6);
    // consecutive waves of the machine take consecutive blocks: neighbours share record cache lines
    uint32_t ub = blockIdx.x * (blockDim.x >> 6) + wave;
    if (ub >= nUB) return;
    UnitStage* S2 = stages + (uint32_t)NBUF * wave;
#ifdef VX_VOX_DEBUG
    unsigned long long vdbg[6] = {0, 0, 0, 0, 0, 0}, vlast = __builtin_readcyclecounter();
#endif
    // triangles [tlo, thi] own the units of pass b (without block_tri: thi - tlo is "too many": global search)
    auto range = [&](uint32_t b, uint32_t& tlo, uint32_t& thi) {
        tlo = 0u;
        thi = 0xFFFFFFF0u;
        if (block_tri) {
            tlo = block_tri[b];
            thi = (b + 1 < nUB) ? block_tri[b + 1] : ntri - 1;
        }
    };
    uint32_t t_lo, t_hi, t_lo_n = 0u, t_hi_n = 0u;
    range(ub, t_lo, t_hi);
    if (t_hi - t_lo < kStageTris) stage_issue(S2[0], recs, unit_base, t_lo, t_hi - t_lo + 1u, lane);
    uint32_t ubn = ub + stride;
    if (ubn < nUB) range(ubn, t_lo_n, t_hi_n);
    int cur = 0;
    for (;;) {
        // this pass's staging (issued one pass ago by this wave) has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VX_V_T(4)
        const UnitStage& S = S2[NBUF == 2 ? cur : 0];
        const uint32_t u = (ub << kUnitBlockLog2) + lane;
        const uint32_t n = t_hi - t_lo + 1u;
        const bool staged = t_hi - t_lo < kStageTris;
        uint32_t tri = 0, rel = 0;
        TriRec r;
        if (u < U) {
            if (staged) {
                uint32_t lo = 0, hi = n;  // S.base[lo] <= u < S.base[hi]
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (S.base[mid] <= u) lo = mid; else hi = mid;
                }
                const float4 a = S.rec[lo * 3], b = S.rec[lo * 3 + 1], c = S.rec[lo * 3 + 2];
                r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
                r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
                r.v[8] = c.x;
                r.xr = __float_as_uint(c.y); r.yr = __float_as_uint(c.z); r.zr = __float_as_uint(c.w);
                rel = u - S.base[lo];
                tri = t_lo + lo;
            } else {
                tri = find_tri(unit_base, ntri, u);
                r = load_rec(recs, tri);
                rel = u - unit_base[tri];
            }
        }
        VX_V_T(1)
        const uint32_t ubnn = ubn + stride;
        uint32_t t_lo_nn = 0u, t_hi_nn = 0u;
        if (ubn < nUB) {
            // (NBUF == 1: into the buffer this pass has just read -- its reads have returned: the waits the compiler puts in front of the
            // first use of their results stand above this point in program order, and LDS operations of a wave complete in order)
            if (NBUF == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (t_hi_n - t_lo_n < kStageTris) stage_issue(S2[NBUF == 2 ? (cur ^ 1) : 0], recs, unit_base, t_lo_n, t_hi_n - t_lo_n + 1u, lane);
            if (ubnn < nUB) range(ubnn, t_lo_nn, t_hi_nn);
        }
        VX_V_T(0)
        f(u, tri, r, rel, u < U);
        VX_V_T(2)
        if (ubn >= nUB) break;
        ub = ubn; ubn = ubnn;
        t_lo = t_lo_n; t_hi = t_hi_n;
        t_lo_n = t_lo_nn; t_hi_n = t_hi_nn;
        cur ^= 1;
    }
#ifdef VX_VOX_DEBUG
    if (lane == 0) for (int i = 0; i < 5; ++i) atomicAdd(&g_vox_dbg[i], vdbg[i]);  // ([5]: requests sent, [6], [7]: k_voxelize)
#endif
}

// ------------------------------------------------------------------------------------------------------------
// K2  voxelize.  One lane per work unit.  The lane sweeps its <=32 voxels along x; everything of the SAT that does
// not depend on x (two box axes, the three e x X axes, n.x) is evaluated once per row and can reject the whole row.
// Hit bits are assembled in a register and leave the lane as at most two atomicOr (one when X % 32 == 0).
// ------------------------------------------------------------------------------------------------------------
#ifdef VX_VOX_FILTER_COHERENT  // experiment: the filter's loads at agent scope (past the XCD's L2)
#define VX_FILTER_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#elif defined(VX_VOX_NO_FILTER)  // experiment: every lane with bits sends its request
#define VX_FILTER_LOAD(p) 0u
#else
#define VX_FILTER_LOAD(p) (*(p))
#endif
#ifndef VX_VOX_NBUF
#define VX_VOX_NBUF 1
#endif
#ifndef VX_VOX_MINWG
#define VX_VOX_MINWG 1
#endif
template <bool EPS, bool STORE_MASK>
__global__ __launch_bounds__(256, VX_VOX_MINWG) void k_voxelize(const TriRec* __restrict__ recs, const uint32_t* __restrict__ unit_base,
                                                  const uint32_t* __restrict__ block_tri, uint32_t ntri, GridParams g, uint32_t* __restrict__ words,
                                                  uint64_t wb, uint64_t we, uint32_t* __restrict__ unit_mask, unsigned long long* set_calls,
                                                  const uint32_t* __restrict__ ext /*null unless the grid has an axis above 65535 cells*/,
                                                  uint32_t* __restrict__ block_hits /*optional, with unit_mask: hits per block of 64 units*/,
                                                  uint32_t tiles_y /*0: `words` is the reference's bitmask; else: the tiled build mask, see k_untile*/, uint32_t xw)
{
    __shared__ UnitStage stage[VX_VOX_NBUF * 4];
    unsigned hits = 0;
#ifdef VX_VOX_DEBUG
    unsigned sent = 0;  // atomic requests this lane really sent
#endif
    for_each_unit<VX_VOX_NBUF>(recs, unit_base, block_tri, ntri, stage, [&](uint32_t u, uint32_t t, const TriRec& r, uint32_t rel, bool valid) {
        uint32_t mask = 0;
        Unit w;
        w.xseg = w.y = w.z = 0u;
        if (valid) {
        w = decode_unit(r, t, rel, ext ? ext[t] : 0u);
        const float cy = cell_centre(g.org[1], g.vs, w.y), cz = cell_centre(g.org[2], g.vs, w.z);
        const SatRow row = sat_row_setup<EPS>(r.v, cy, cz, g.half);
#ifdef VX_DIAG_NO_SAT
        if (row.alive && w.x0 == 0xFFFFFFFFu) {
#else
        if (row.alive) {
#endif
            // (a two-sweep form -- box x + plane on every voxel, the six edge axes on the survivors only -- is slower: the
            // wave pays for its lane with the most survivors, 0.21 ms vs 0.18 ms; unrolled by two, the sweep needs twice the
            // registers for no gain)
            // The x range is K2a's, trimmed with the box-axis test along x: the sweep leaves that test out (sat_row_test, BOXX).  (x + 0.5f
            // is carried as a float: exact below 2^23, one conversion less per voxel.)
            float xh = (float)w.x0 + 0.5f;
            uint32_t bit = 1u << (w.x0 & 31u);
            // EPS variant: when no lane's row can have an axis shorter than the threshold, the sweep without those checks.
            // (Two voxels per turn on two-component vectors -- v_pk_mul_f32 / v_pk_add_f32 -- measured: 178 / 156 VALU instructions per PAIR
            // against 132 / 117 per voxel, but 125 VGPRs and 91 us against 68: the packed forms are no faster per float here.)
            if (EPS && !__any(!sat_row_all_live(row))) {
#pragma clang loop unroll(disable) vectorize(disable)
                for (uint32_t x = w.x0; x < w.x1; ++x) {
                    const float cx = g.org[0] + (xh * g.vs);  // cell_centre
                    if (sat_row_test<EPS, false, false>(row, r.v, cx, g.half)) mask |= bit;
                    xh += 1.0f;
                    bit <<= 1;
                }
            } else {
#pragma clang loop unroll(disable) vectorize(disable)
                for (uint32_t x = w.x0; x < w.x1; ++x) {
                    const float cx = g.org[0] + (xh * g.vs);  // cell_centre
                    if (sat_row_test<EPS, false>(row, r.v, cx, g.half)) mask |= bit;
                    xh += 1.0f;
                    bit <<= 1;
                }
            }
        }
        }
        if (STORE_MASK) {
            // the hits of the block's 64 units: what the ordered emission scans instead of the 64 masks (all lanes are here)
            if (valid) unit_mask[u] = mask;
            if (block_hits) {
                const uint32_t tot = wave_sum_u32(__popc(mask));
                if ((threadIdx.x & 63u) == 0u) block_hits[u >> kUnitBlockLog2] = tot;
            }
        }
#ifdef VX_VOX_DEBUG  // [6] lanes of a wave pass that have bits to set, [7] distinct first words among them
        {
            const unsigned long long act = __ballot(mask != 0);
            const uint32_t key = (uint32_t)(((uint64_t)g.dim[0] * ((uint64_t)w.y + (uint64_t)g.dim[1] * w.z) + w.xseg) >> 5);
            unsigned distinct = 0;
            unsigned long long rem = act;
            while (rem) {
                const int l = __ffsll((long long)rem) - 1;
                const uint32_t k = __shfl(key, l, 64);
                rem &= ~__ballot(key == k);
                ++distinct;
            }
            if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) {
                atomicAdd(&g_vox_dbg[6], (unsigned long long)__popcll(act));
                atomicAdd(&g_vox_dbg[7], (unsigned long long)distinct);
            }
        }
#endif
        if (mask) {
            uint64_t wi;
            uint32_t lo, hi;
            if (tiles_y) {  // tiled build mask (rows are whole words): the unit's word, next to those of its y and z neighbours
                const uint64_t tile = ((uint64_t)(w.z >> 2) * tiles_y + (w.y >> 2)) * xw + (w.xseg >> 5);
                wi = tile * 16ull + ((w.z & 3u) << 2) + (w.y & 3u);
                lo = mask;
                hi = 0u;
            } else {
                const uint64_t i0 = (uint64_t)g.dim[0] * ((uint64_t)w.y + (uint64_t)g.dim[1] * w.z) + w.xseg;  // map3dto1d, voxelgrid.hpp:37-40
                const uint32_t sh = (uint32_t)i0 & 31u;
                wi = i0 >> 5;
                lo = mask << sh;
                hi = sh ? (mask >> (32u - sh)) : 0u;
            }
            // The atomics execute at the memory side, one 64-byte request per lane whatever the wave's address pattern: their
            // request rate (about 20 G/s for the chip), not the SAT, bounds this kernel.
#ifdef VX_DIAG_NO_ATOMICS  // diagnostic build: price of the atomics (results are wrong)
            hits += __popc(lo) + __popc(hi);
#elif defined(VX_DIAG_PLAIN_STORE)  // diagnostic build: the same addresses by plain stores (results are wrong)
            if (lo && wi >= wb && wi < we) { words[wi] = lo; hits += __popc(lo); }
            if (hi && wi + 1 >= wb && wi + 1 < we) { words[wi + 1] = hi; hits += __popc(hi); }
#else
            // A plain load first: when it already shows every bit of the lane's mask set, the request is not sent (-10 % of the
            // kernel on the bench scene).  The copy it reads may be stale -- this XCD's L2 or the CU's L1 filled earlier in this
            // kernel -- but never wrong in the unsafe direction: bits only get set during a build, the atomics execute at the
            // memory side and drop the line from L2 instead of updating it, nothing else writes the mask in this kernel, and
            // what older kernels left in the caches (the previous build's mask) is invalidated at the kernel boundary like every
            // other buffer this pipeline passes from kernel to kernel.  A stale copy can only show fewer bits: one request more.
            // A mask written from OUTSIDE the library between two builds (vx_grid_bitmask_device_mut: the RCCL all-gather / the peer
            // copies of a multi-rank exchange) is covered the same way: every build clears its mask first -- k_tri_setup's threads or
            // a memset, both kernels in front of this one on the stream -- so the loads here only ever see this build's own bits
            // (tests/test_gpu_parity.py::test_rebuild_after_external_write_of_the_mask).
            // (The tiled build mask needs no filter: its requests are few enough for the memory side, and without the loads the wave does
            // not wait for anything between its sweep and its next pass: 71.3 -> 69.2 us.)
            if (tiles_y) {
                // (a word shard: only the rows whose word of the reference's mask lies in [wb, we) -- the z slab of a shard may begin and
                // end inside a plane)
                const uint64_t lin = (uint64_t)xw * ((uint64_t)w.y + (uint64_t)g.dim[1] * w.z) + (w.xseg >> 5);
                if (lin >= wb && lin < we) {
                    atomicOr(&words[wi], lo);  // voxelgridBool.cpp:66
                    VX_V_SENT
                    hits += __popc(lo);
                }
            } else {
                if (lo && wi >= wb && wi < we) { if ((VX_FILTER_LOAD(&words[wi]) & lo) != lo) { atomicOr(&words[wi], lo); VX_V_SENT } hits += __popc(lo); }        // voxelgridBool.cpp:66
                if (hi && wi + 1 >= wb && wi + 1 < we) { if ((VX_FILTER_LOAD(&words[wi + 1]) & hi) != hi) { atomicOr(&words[wi + 1], hi); VX_V_SENT } hits += __popc(hi); }
            }
#endif
        }
    });
    hits = wave_sum_u32(hits);
#ifdef VX_VOX_DEBUG
    sent = wave_sum_u32(sent);
    if ((threadIdx.x & 63) == 0 && sent) atomicAdd(&g_vox_dbg[5], (unsigned long long)sent);
#endif
#ifndef VX_DIAG_NO_SETCALLS
    // 64 counters on 64-byte lines of their own, a wave adds to one of them: thousands of adds on ONE address queue at the memory-side
    // atomic unit at ~12 ns each -- 8192 waves: 100 us, which was this kernel's tail once the bitmask requests got fewer
    if ((threadIdx.x & 63) == 0 && hits) atomicAdd(set_calls + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (kCallCounters - 1u)) * 8u, (unsigned long long)hits);
#endif
}

#ifdef VX_VOX_DEBUG
}  // namespace vx
extern "C" int vx_debug_vox(unsigned long long* out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(vx::g_vox_dbg), 8 * 8) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(vx::g_vox_dbg), z, 8 * 8) != hipSuccess) return 1; }
    return 0;
}
namespace vx {
#endif

void launch_voxelize(const TriRec* recs, const uint32_t* unit_base, const uint32_t* block_tri, uint32_t ntri, const GridParams& g, int sat_variant,
                     uint32_t* words, uint64_t wb, uint64_t we, uint32_t* unit_mask, unsigned long long* set_calls, hipStream_t s, const uint32_t* ext,
                     uint32_t* block_hits, bool tiled)
{
    if (!ntri) return;
    const dim3 grid(kUnitBlocks), block(256);
    const uint32_t ty = tiled ? (g.dim[1] + 3u) / 4u : 0u, xw = g.dim[0] / 32u;
    if (sat_variant == 0) {
        if (unit_mask) VX_KL((k_voxelize<true, true>), grid, block, 0, s, recs, unit_base, block_tri, ntri, g, words, wb, we, unit_mask, set_calls, ext, block_hits, ty, xw);
        else VX_KL((k_voxelize<true, false>), grid, block, 0, s, recs, unit_base, block_tri, ntri, g, words, wb, we, unit_mask, set_calls, ext, (uint32_t*)nullptr, ty, xw);
    } else {
        if (unit_mask) VX_KL((k_voxelize<false, true>), grid, block, 0, s, recs, unit_base, block_tri, ntri, g, words, wb, we, unit_mask, set_calls, ext, block_hits, ty, xw);
        else VX_KL((k_voxelize<false, false>), grid, block, 0, s, recs, unit_base, block_tri, ntri, g, words, wb, we, unit_mask, set_calls, ext, (uint32_t*)nullptr, ty, xw);
    }
}

// ------------------------------------------------------------------------------------------------------------
// The tiled build mask.  The memory-side atomic unit takes one request per 64-byte line and wave instruction, and it is its request
// rate that bounds K2.  In the reference's layout (x fastest) the words of a triangle's rows lie a row apart -- one request per
// lane.  For grids whose rows are whole words (X % 32 == 0) an unsharded build therefore ORs into a mask in which the sixteen words
// (32 voxels along x each) of 4 x 4 rows (y, z) share a line:
//     word(xs, y, z) = (((z / 4) * ceil(Y / 4) + y / 4) * (X / 32) + xs) * 16 + (z % 4) * 4 + y % 4
// -- consecutive lanes of a pass are consecutive rows of one triangle, so a triangle sends about three requests instead of ten --
// and k_untile writes the reference's bitmask from it (through LDS, 16-byte accesses on both sides).
// ------------------------------------------------------------------------------------------------------------
uint64_t tiled_mask_words(const uint32_t dim[3]) { return (uint64_t)((dim[2] + 3u) / 4u) * ((dim[1] + 3u) / 4u) * (dim[0] / 32u) * 16ull; }

// One workgroup = four tile rows (ty, tz) x sixteen tiles along x = 4 KiB: one 16-byte load per thread (the four y words of one z of a
// tile), sixteen-by-sixteen transposes in LDS, one 16-byte store per thread (four consecutive words of a row of the reference's mask).
// [wb, we): the words this build owns (a word shard; the whole mask otherwise) -- every OTHER word of the mask is written as zero, as a
// sharded build leaves it (the exchange fills those in).
__global__ __launch_bounds__(256) void k_untile(const uint32_t* __restrict__ tiled, uint32_t* __restrict__ words, uint32_t xw, uint32_t Y, uint32_t Z, uint32_t tiles_y,
                                                uint32_t ntile_rows /*tiles_y * ceil(Z / 4)*/, uint32_t xchunks /*ceil(xw / 16)*/, uint64_t wb, uint64_t we)
{
    __shared__ uint32_t sh[4][16][17];  // [tile row of the group][z % 4 * 4 + y % 4][tile], padded
    const uint32_t q = threadIdx.x >> 6, l = threadIdx.x & 63u;
    const uint32_t grp = blockIdx.x / xchunks, x0 = (blockIdx.x - grp * xchunks) * 16u;
    const uint32_t tr = grp * 4u + q;
    {
        const uint32_t xt = l >> 2, zz = l & 3u;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (tr < ntile_rows && x0 + xt < xw) v = *reinterpret_cast<const uint4*>(tiled + ((uint64_t)tr * xw + x0 + xt) * 16ull + zz * 4u);
        sh[q][zz * 4u + 0u][xt] = v.x; sh[q][zz * 4u + 1u][xt] = v.y; sh[q][zz * 4u + 2u][xt] = v.z; sh[q][zz * 4u + 3u][xt] = v.w;
    }
    __syncthreads();
    if (tr >= ntile_rows) return;
    const uint32_t tz = tr / tiles_y, ty = tr - tz * tiles_y;
    const uint32_t row = l >> 2, xs = x0 + (l & 3u) * 4u;
    const uint32_t y = ty * 4u + (row & 3u), z = tz * 4u + (row >> 2);
    if (y >= Y || z >= Z || xs >= xw) return;
    const uint64_t d0 = (uint64_t)xw * ((uint64_t)y + (uint64_t)Y * z) + xs;
    uint32_t* dst = words + d0;
    const uint32_t* src = &sh[q][row][(l & 3u) * 4u];
    if ((xw & 3u) == 0u && d0 >= wb && d0 + 4u <= we) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(src[0], src[1], src[2], src[3]);
    } else if ((xw & 3u) == 0u && (d0 + 4u <= wb || d0 >= we)) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (uint32_t k = 0; k < 4u && xs + k < xw; ++k) dst[k] = (d0 + k >= wb && d0 + k < we) ? src[k] : 0u;
    }
}

void launch_untile(const uint32_t* tiled, uint32_t* words, const uint32_t dim[3], hipStream_t s, uint64_t wb, uint64_t we)
{
    const uint32_t ty = (dim[1] + 3u) / 4u, tz = (dim[2] + 3u) / 4u, xw = dim[0] / 32u;
    const uint64_t rows = (uint64_t)ty * tz;
    if (!rows || !xw) return;
    const uint32_t xchunks = (xw + 15u) / 16u;
    const uint64_t nblk = ((rows + 3ull) / 4ull) * xchunks;  // (at most 2^37 / 32 / 16 / 4 workgroups: fits the grid's 2^31)
    VX_KL(k_untile, dim3((unsigned)nblk), dim3(256), 0, s, tiled, words, xw, dim[1], dim[2], ty, (uint32_t)rows, xchunks, wb, we);
}

// ------------------------------------------------------------------------------------------------------------
// K3  ordered emission with duplicates.  Work units are laid out triangle-major, then z, y, x: exactly the order
// of the reference's loop nest (VoxelBuilder.hpp:186-195), so the exclusive scan of the units' hit counts is the
// position of a unit's first hit in VoxelGridVec::m_voxel / in the octree's pre-sort item list.
//
// The scan is two-level: K2 leaves the hits of every block of 64 units (one wave pass), the device scan runs over those
// (U / 64 values instead of U), and the wave that emits a block scans its 64 popcounts itself.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += o;
    }
    return v;
}

__device__ __forceinline__ uint32_t select_bit(uint32_t v, uint32_t r);

__global__ __launch_bounds__(256) void k_emit_units(const TriRec* __restrict__ recs, const uint32_t* __restrict__ unit_base,
                                                    const uint32_t* __restrict__ block_tri, uint32_t ntri, GridParams g,
                                                    const uint32_t* __restrict__ unit_mask, const uint32_t* __restrict__ block_base /*exclusive scan of K2's block hits*/,
                                                    vx_aabb* __restrict__ aabbs, uint64_t* __restrict__ morton, uint64_t cap /*records the output can hold*/,
                                                    const uint32_t* __restrict__ ext)
{
    // One staging buffer per wave: 13 KiB of LDS per workgroup, eight workgroups (every wave slot) per CU instead of six.  This kernel waits
    // for its own stores -- a wave's memory operations retire in issue order, so the staging of its next pass waits for the records of this
    // one -- and more waves in flight are what hides that.
    __shared__ UnitStage stage[VX_EMIT_NBUF * 4];
    const uint32_t lane = threadIdx.x & 63u;
    for_each_unit<VX_EMIT_NBUF>(recs, unit_base, block_tri, ntri, stage, [&](uint32_t u, uint32_t t, const TriRec& r, uint32_t rel, bool valid) {
        const uint32_t mask = valid ? unit_mask[u] : 0u;
        const uint32_t cnt = __popc(mask);
        const uint32_t incl = wave_incl_scan_u32(cnt, lane);
        const uint32_t excl = incl - cnt;
        const uint32_t T = __shfl(incl, 63, 64);  // records of this block
        if (!T) return;
        const uint64_t base = block_base[u >> kUnitBlockLog2];
#ifndef VX_EMIT_OUTPUT_CENTRIC
        if (!mask) return;
        const Unit w = decode_unit(r, t, rel, ext ? ext[t] : 0u);
        uint64_t off = base + excl;
        uint32_t m = mask;
        while (m) {
            const uint32_t b = __ffs(m) - 1;
            m &= m - 1;
            const uint32_t x = w.xseg + b;
            if (off >= cap) return;  // speculative emission into an existing buffer: the host re-emits after growing it
            if (aabbs) {
                float bb[6];
                cell_aabb(g, x, w.y, w.z, bb);
                float2* o = reinterpret_cast<float2*>(aabbs + off);
                o[0] = make_float2(bb[0], bb[1]);
                o[1] = make_float2(bb[2], bb[3]);
                o[2] = make_float2(bb[4], bb[5]);
            }
            if (morton) morton[off] = morton3d(x, w.y, w.z);  // octTree.hpp:765
            ++off;
        }
#else
        // OUTPUT-centric form (measured, not the default): record j of the block's contiguous output range goes to lane j % 64, which
        // finds the unit that owns it (binary search of the 64 exclusive counts, by lane permutes) and the bit inside that unit's mask --
        // consecutive lanes write consecutive 24-byte records whatever the units' hit counts.  42.7 us against 39 on the bench scene:
        // the eleven permutes per 64 records cost more than the tidier stores give.
        uint32_t xseg = 0u, y = 0u, z = 0u;
        if (mask) {
            const Unit w = decode_unit(r, t, rel, ext ? ext[t] : 0u);
            xseg = w.xseg; y = w.y; z = w.z;
        }
        for (uint32_t j0 = 0; j0 < T; j0 += 64u) {
            const uint32_t j = j0 + lane;
            // owner of record j: the largest lane l with excl[l] <= j (units without hits share their count with the next one and lose)
            uint32_t lo = 0u, hi = 64u;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const uint32_t mid = (lo + hi) >> 1;
                const uint32_t e = __shfl(excl, (int)mid, 64);
                if (e <= j) lo = mid; else hi = mid;
            }
            const uint32_t oe = __shfl(excl, (int)lo, 64), om = __shfl(mask, (int)lo, 64);
            const uint32_t ox = __shfl(xseg, (int)lo, 64), oy = __shfl(y, (int)lo, 64), oz = __shfl(z, (int)lo, 64);
            const uint64_t off = base + j;
            if (j >= T || off >= cap) continue;  // (speculative emission into an existing buffer: the host re-emits after growing it)
            const uint32_t x = ox + select_bit(om, j - oe);
            if (aabbs) {
                float bb[6];
                cell_aabb(g, x, oy, oz, bb);
                float2* o = reinterpret_cast<float2*>(aabbs + off);
                o[0] = make_float2(bb[0], bb[1]);
                o[1] = make_float2(bb[2], bb[3]);
                o[2] = make_float2(bb[4], bb[5]);
            }
            if (morton) morton[off] = morton3d(x, oy, oz);  // octTree.hpp:765
        }
#endif
    });
}

void launch_emit_units(const TriRec* recs, const uint32_t* unit_base, const uint32_t* block_tri, uint32_t ntri, const GridParams& g,
                       const uint32_t* unit_mask, const uint32_t* block_base, vx_aabb* aabbs, uint64_t* morton, hipStream_t s, uint64_t cap, const uint32_t* ext)
{
    if (!ntri) return;
    // (bound by its stores, not by issue: 4096 workgroups -- several ragged rounds of the six a CU holds -- beat one even set: 45 us
    // with 1536, 41 with 2048, 39 with 4096)
    VX_KL(k_emit_units, dim3(VX_EMIT_BLOCKS), dim3(256), 0, s, recs, unit_base, block_tri, ntri, g, unit_mask, block_base, aabbs, morton, cap, ext);
}

// ------------------------------------------------------------------------------------------------------------
// K4  VoxelGridBool::getAabbs: ascending word, ascending bit (voxelgridBool.cpp:18-52).  Output-centric AND output-balanced:
// a workgroup owns 1024 consecutive OUTPUT records (not a slab of the bitmask: a fully occupied wall would give one
// workgroup 32x the work of its neighbours -- measured 150 us with input tiles).  It locates the words that hold them with
// one binary search of word_prefix, stages that word range in LDS when it is short (dense regions) or searches
// word_prefix globally inside the range (sparse regions), selects the k-th set bit, and writes the records as one
// contiguous run of 8-byte pieces through LDS: 4 B/word in, 24 B/occupied voxel out.
// word_prefix[w] = number of set bits before word w (nwords + 1 entries).
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kEmitOut = 1024;    // output records per workgroup
constexpr uint32_t kEmitStage = 2048;  // bitmask words staged in LDS

__device__ __forceinline__ uint32_t select_bit(uint32_t v, uint32_t r)  // position of the r-th (0-based) set bit of v
{
    uint32_t pos = 0, c;
    c = __popc(v & 0xFFFFu); if (r >= c) { r -= c; pos += 16; v >>= 16; }
    c = __popc(v & 0xFFu);   if (r >= c) { r -= c; pos += 8;  v >>= 8; }
    c = __popc(v & 0xFu);    if (r >= c) { r -= c; pos += 4;  v >>= 4; }
    c = __popc(v & 0x3u);    if (r >= c) { r -= c; pos += 2;  v >>= 2; }
    c = v & 1u;              if (r >= c) { pos += 1; }
    return pos;
}

// largest w in [lo, hi) with pre[w] <= k   (pre non-decreasing, pre[lo] <= k < pre[hi])
__device__ __forceinline__ uint64_t upper_word(const uint32_t* __restrict__ pre, uint64_t lo, uint64_t hi, uint32_t k)
{
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pre[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}

// `sel` (optional, written by the prefix scan): sel[c] = the word that holds output record 1024 c -- without it every workgroup finds
// its word range by two binary searches of the prefix array in global memory: 44 dependent loads, 30 of the kernel's 52 us.
__global__ __launch_bounds__(256) void k_emit_bool(const uint32_t* __restrict__ words, const uint32_t* __restrict__ word_prefix, GridParams g,
                                                   vx_aabb* __restrict__ out, uint64_t capacity, const uint32_t* __restrict__ sel)
{
    __shared__ uint32_t s_pre[kEmitStage + 1];
    __shared__ uint32_t s_word[kEmitStage];
    __shared__ __attribute__((aligned(16))) float s_rec[256 * 6];
    const uint32_t total = word_prefix[g.nwords];
    const uint64_t XY = (uint64_t)g.dim[0] * g.dim[1];
    const uint64_t limit = total < capacity ? total : capacity;
    for (uint64_t ob = (uint64_t)blockIdx.x * kEmitOut; ob < limit; ob += (uint64_t)gridDim.x * kEmitOut) {
        const uint32_t o_first = (uint32_t)ob;
        const uint32_t o_last = (uint32_t)((ob + kEmitOut < limit ? ob + kEmitOut : limit) - 1);
        // word range of this output range (uniform: every lane runs the same search, the loads are broadcast)
        static_assert(kEmitOut == 1024, "sel is sampled every 1024 records");
        const uint64_t w_lo = sel ? (uint64_t)sel[ob >> 10] : upper_word(word_prefix, 0, g.nwords, o_first);
        // a full chunk ends where the next one begins: the word that holds record o_last + 1 is at or behind the one that holds o_last
        const bool next_known = sel && (uint64_t)o_last + 1 == ob + kEmitOut && (uint64_t)o_last + 1 < total;
        const uint64_t w_hi = next_known ? (uint64_t)sel[(ob >> 10) + 1] : upper_word(word_prefix, w_lo, g.nwords, o_last);
        const uint32_t nw = (uint32_t)(w_hi - w_lo + 1);
        const bool staged = nw <= kEmitStage;
        if (staged) {
            for (uint32_t i = threadIdx.x; i <= nw; i += 256u) s_pre[i] = word_prefix[w_lo + i];
            for (uint32_t i = threadIdx.x; i < nw; i += 256u) s_word[i] = words[w_lo + i];
            __syncthreads();
        }
        for (uint32_t kb = o_first; kb <= o_last; kb += 256u) {
            const uint32_t k = kb + threadIdx.x;
            if (k <= o_last) {
                uint64_t w;
                uint32_t wv, before;
                if (staged) {
                    uint32_t lo = 0, hi = nw;  // s_pre[lo] <= k < s_pre[hi]
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_pre[mid] <= k) lo = mid; else hi = mid;
                    }
                    w = w_lo + lo; wv = s_word[lo]; before = s_pre[lo];
                } else {
                    w = upper_word(word_prefix, w_lo, w_hi + 1, k);
                    wv = words[w]; before = word_prefix[w];
                }
                const uint32_t bit = select_bit(wv, k - before);
                const uint64_t idx = w * 32ull + bit;  // voxel index -> (x, y, z), voxelgrid.hpp:42-49
                const uint32_t z = (uint32_t)(idx / XY);
                const uint64_t rem = idx - (uint64_t)z * XY;  // (X * Y may exceed 32 bits on a long thin grid)
                const uint32_t y = (rem >> 32) ? (uint32_t)(rem / g.dim[0]) : (uint32_t)rem / g.dim[0];
                const uint32_t x = (uint32_t)(rem - (uint64_t)y * g.dim[0]);
                float bb[6];
                cell_aabb(g, x, y, z, bb);
                float2* sp = reinterpret_cast<float2*>(s_rec) + threadIdx.x * 3u;
                sp[0] = make_float2(bb[0], bb[1]);
                sp[1] = make_float2(bb[2], bb[3]);
                sp[2] = make_float2(bb[4], bb[5]);
            }
            __syncthreads();
            // the chunk's records leave as one contiguous run of 8-byte pieces: consecutive lanes, consecutive addresses
            const uint32_t nrec = (o_last - kb + 1) < 256u ? (o_last - kb + 1) : 256u;
            float2* gp = reinterpret_cast<float2*>(out + kb);
            const float2* sp2 = reinterpret_cast<const float2*>(s_rec);
            for (uint32_t i = threadIdx.x; i < nrec * 3u; i += 256u) gp[i] = sp2[i];
            __syncthreads();
        }
    }
}

void launch_emit_bool_aabbs(const uint32_t* words, const uint32_t* word_prefix, const GridParams& g, vx_aabb* out, uint64_t capacity,
                            hipStream_t s, const uint32_t* sel1024)
{
    if (!g.nwords || !capacity) return;
    // the grid is sized by the caller's capacity (the count is only known on the device); surplus workgroups exit at once
    const uint64_t nblk = (capacity + kEmitOut - 1) / kEmitOut;
    VX_KL(k_emit_bool, dim3((unsigned)(nblk < 65536 ? nblk : 65536)), dim3(256), 0, s, words, word_prefix, g, out, capacity, sel1024);
}

// Octree::getAabbs: DFS over the node array visits items in sorted order (octTree.hpp:374-392); decode + AABB per item.
__global__ __launch_bounds__(256) void k_emit_morton_aabbs(const uint64_t* __restrict__ items, uint64_t n, float ox, float oy, float oz,
                                                           float vs, vx_aabb* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const uint64_t m = items[i];
        GridParams g;
        g.org[0] = ox; g.org[1] = oy; g.org[2] = oz; g.vs = vs; g.half = vs * 0.5f;
        float bb[6];
        cell_aabb(g, compact_bits(m), compact_bits(m >> 1), compact_bits(m >> 2), bb);
        float2* o = reinterpret_cast<float2*>(out + i);
        o[0] = make_float2(bb[0], bb[1]);
        o[1] = make_float2(bb[2], bb[3]);
        o[2] = make_float2(bb[4], bb[5]);
    }
}
void launch_emit_morton_aabbs(const uint64_t* items, uint64_t n, const float root_min[3], float vs, vx_aabb* out, hipStream_t s)
{
    if (!n) return;
    VX_KL(k_emit_morton_aabbs, dim3(grid_for(n, 256, kMaxBlocks)), dim3(256), 0, s, items, n, root_min[0], root_min[1],
                       root_min[2], vs, out);
}

// ------------------------------------------------------------------------------------------------------------
// Per-voxel material ids: the plumbing the reference keeps commented out (VoxelBuilder.hpp:375-395 -> setVoxel ->
// addMatrialIfNeeded, voxelgrid.hpp:102-114, call sites voxelgridBool.cpp:64, voxelgridAABBstruct.cpp:31,
// voxelgridVecEncoding.cpp:27).  m_matIdx[idx] is overwritten by every setVoxel call on the voxel, so what survives is the
// material of the LAST call = of the highest-numbered triangle that hits the voxel (calls run in triangle order and a triangle
// sets a voxel at most once).  Pass 1 takes that maximum per occupied voxel (addressed by its rank in the ascending AABB
// list) and marks the triangles that hit anything (the order in which materials are first used defines their index);
// pass 2 maps triangle -> material value -> index.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mat_last(const TriRec* __restrict__ recs, const uint32_t* __restrict__ unit_base, const uint32_t* __restrict__ block_tri,
                                                  uint32_t ntri, GridParams g, const uint32_t* __restrict__ unit_mask, const uint32_t* __restrict__ words,
                                                  const uint32_t* __restrict__ word_prefix, uint32_t* __restrict__ last_tri /*per occupied voxel, 0 = none yet*/,
                                                  uint8_t* __restrict__ tri_hit, const uint32_t* __restrict__ ext)
{
    __shared__ UnitStage stage[kStagesPerBlock];
    for_each_unit(recs, unit_base, block_tri, ntri, stage, [&](uint32_t u, uint32_t t, const TriRec& r, uint32_t rel, bool valid) {
        uint32_t mask = valid ? unit_mask[u] : 0u;
        if (!mask) return;
        tri_hit[t] = 1;
        if (!last_tri) return;
        const Unit w = decode_unit(r, t, rel, ext ? ext[t] : 0u);
        const uint64_t row = (uint64_t)g.dim[0] * ((uint64_t)w.y + (uint64_t)g.dim[1] * w.z);
        while (mask) {
            const uint32_t b = __ffs(mask) - 1;
            mask &= mask - 1;
            const uint64_t i = row + w.xseg + b;
            const uint64_t wi = i >> 5;
            const uint32_t wd = words[wi];
            if (!((wd >> (i & 31u)) & 1u)) continue;  // a voxel of the z slab outside this build's word shard: not in this mask, not ranked here
            const uint32_t rank = word_prefix[wi] + __popc(wd & ((1u << (i & 31u)) - 1u));
            atomicMax(&last_tri[rank], t + 1u);
        }
    });
}

// Bool / AABBstruct: one id per occupied voxel, in ascending voxel order (== the order of getAabbs and of VoxelGrid::getMatIdx)
__global__ __launch_bounds__(256) void k_mat_ids(const uint32_t* __restrict__ last_tri, uint64_t n, const int32_t* __restrict__ tri_value,
                                                 const int16_t* __restrict__ value_index, int16_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t lt = last_tri[i];
        out[i] = lt ? value_index[tri_value[lt - 1u]] : (int16_t)-1;
    }
}

// Vec: addMatrialIfNeeded(m_voxelSet, material) -- one id per setVoxel call, in call order (the order of the Aabb list)
__global__ __launch_bounds__(256) void k_mat_ids_calls(const TriRec* __restrict__ recs, const uint32_t* __restrict__ unit_base, const uint32_t* __restrict__ block_tri,
                                                       uint32_t ntri, const uint32_t* __restrict__ unit_mask, const uint32_t* __restrict__ block_base /*as in k_emit_units*/,
                                                       const int32_t* __restrict__ tri_value, const int16_t* __restrict__ value_index, int16_t* __restrict__ out)
{
    __shared__ UnitStage stage[kStagesPerBlock];
    for_each_unit(recs, unit_base, block_tri, ntri, stage, [&](uint32_t u, uint32_t t, const TriRec&, uint32_t, bool valid) {
        const uint32_t n = valid ? __popc(unit_mask[u]) : 0u;
        const uint32_t incl = wave_incl_scan_u32(n, threadIdx.x & 63u);  // (all lanes)
        if (!n) return;
        const int16_t id = value_index[tri_value[t]];
        const uint32_t off = block_base[u >> kUnitBlockLog2] + (incl - n);
        for (uint32_t k = 0; k < n; ++k) out[off + k] = id;
    });
}

void launch_mat_last(const TriRec* recs, const uint32_t* unit_base, const uint32_t* block_tri, uint32_t ntri, const GridParams& g, const uint32_t* unit_mask,
                     const uint32_t* words, const uint32_t* word_prefix, uint32_t* last_tri, uint8_t* tri_hit, hipStream_t s, const uint32_t* ext)
{
    if (!ntri) return;
    VX_KL(k_mat_last, dim3(kMaxBlocks), dim3(256), 0, s, recs, unit_base, block_tri, ntri, g, unit_mask, words, word_prefix, last_tri, tri_hit, ext);
}
void launch_mat_ids(const uint32_t* last_tri, uint64_t n, const int32_t* tri_value, const int16_t* value_index, int16_t* out, hipStream_t s)
{
    if (!n) return;
    VX_KL(k_mat_ids, dim3(grid_for(n, 256, kMaxBlocks)), dim3(256), 0, s, last_tri, n, tri_value, value_index, out);
}
void launch_mat_ids_calls(const TriRec* recs, const uint32_t* unit_base, const uint32_t* block_tri, uint32_t ntri, const uint32_t* unit_mask,
                          const uint32_t* block_base, const int32_t* tri_value, const int16_t* value_index, int16_t* out, hipStream_t s)
{
    if (!ntri) return;
    VX_KL(k_mat_ids_calls, dim3(kMaxBlocks), dim3(256), 0, s, recs, unit_base, block_tri, ntri, unit_mask, block_base, tri_value, value_index, out);
}

__global__ void k_set_bit(uint32_t* words, uint64_t idx) { atomicOr(&words[idx >> 5], 1u << (idx & 31)); }
void launch_set_bit(uint32_t* words, uint64_t idx, hipStream_t s) { VX_KL(k_set_bit, dim3(1), dim3(1), 0, s, words, idx); }

}  // namespace vx
