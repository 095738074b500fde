// vx_octree.hip -- device construction of the octree's flat node array from the sorted Morton items
// (replaces Octree::buildNodeRecursive, octTree.hpp:319-358).
//
// The reference recurses depth-first: node = {children[8], start, count}, leaf iff depth >= maxDepth || count <= maxItems,
// child octant = (code >> 3*(maxDepth-1-depth)) & 7, nodes appended in PRE-ORDER.
//
// DIRECT FORM (maxItems <= 64, the reference's default is 16) -- no recursion, no levels, one host wait for the node count:
//   * a node is a maximal run of items sharing their first d octal digits (its depth); in pre-order a node precedes exactly the
//     nodes with a larger (start, depth) pair, so the nodes that START at item position i are consecutive in the array, ordered by
//     depth: depths d0(i) .. d1(i);
//   * d0(i) = lcp(item[i-1], item[i]) + 1 (0 for i = 0): a run of every deeper depth begins where the common prefix ends;
//   * a node of depth d exists iff its parent splits, i.e. iff the depth-(d-1) run holding i has more than maxItems items.  Runs
//     are contiguous, so that run has > m items iff some window item[j], item[j+m] with j <= i <= j+m shares d-1 digits:
//     D(i) = max over j in [i-m, i] of lcp(item[j], item[j+m]) is the deepest splitting depth over i, d1(i) = min(maxDepth, D(i)+1);
//   * k_oct_depths writes n(i) = d1 - d0 + 1 (or 0) per position, an exclusive scan gives the pre-order index of (i, d0(i));
//   * k_oct_nodes: thread i writes start / count of its nodes (the end of a run by a galloping search from i) and links each node
//     into its parent -- the previous node of the same thread, or for the shallowest one the node that owns the run of depth
//     d0-1 around i (start found by a backward galloping search).  The node array is pre-filled with 0xFF (children = none).
// Traffic: items twice, one byte and one uint32 per item, 40 B per node -- against a sort of (start, depth) keys and two
// passes of seven binary searches per node and level in the level-by-level form below (kept for maxItems > 64).
//
// LEVEL-BY-LEVEL FORM: the tree is expanded breadth-first,
// one level per pass (every node of a level splits its item range with seven binary searches), and the pre-order numbering
// is recovered afterwards: in pre-order a node precedes exactly the nodes with a larger (start, depth) pair -- a node's
// subtree is a contiguous item range, children are visited in ascending start, and nested nodes with equal start are
// visited shallower first -- so the pre-order index is the rank of the key (start << 8 | depth), obtained with one radix
// sort of (key, breadth-first id) pairs; child links are then remapped through the inverse permutation.
#include "vx_internal.h"

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace vx {

#define VX_KL(kern, grid, block, shmem, stream, ...)                         \
    do {                                                                     \
        ProfScope ps_(#kern, stream);                                        \
        hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);   \
    } while (0)

namespace {

// number of leading octal digits (of `bits` per code) two codes share
__device__ __forceinline__ int lcp_digits(uint64_t a, uint64_t b, int bits)
{
    if (a == b) return bits;
    const int hb = 63 - __clzll((long long)(a ^ b));  // highest differing bit
    const int d = (3 * bits - 1 - hb) / 3;            // digits above it
    return d < 0 ? 0 : d;
}

constexpr uint32_t kDepthTile = 1024;  // item positions per workgroup
constexpr uint32_t kDepthHalo = (uint32_t)kOctDirectMaxItems;

__global__ __launch_bounds__(256) void k_oct_depths(const uint64_t* __restrict__ items, uint32_t n, int bits, uint32_t m, uint8_t* __restrict__ ncount)
{
    __shared__ uint64_t codes[kDepthTile + 2 * kDepthHalo + 2];  // codes[c] = item[t0 - (m + 1) + c]
    __shared__ int8_t L[kDepthTile + kDepthHalo];                // L[c] = lcp(item[j], item[j + m]) for j = t0 - m + c (-1: no such window)
    const int64_t t0 = (int64_t)blockIdx.x * kDepthTile;
    const int64_t lo = t0 - (int64_t)(m + 1);
    const uint32_t ncodes = kDepthTile + 2u * m + 1u;
    for (uint32_t c = threadIdx.x; c < ncodes; c += 256u) {
        const int64_t j = lo + c;
        codes[c] = (j >= 0 && j < (int64_t)n) ? items[j] : 0ull;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < kDepthTile + m; c += 256u) {
        const int64_t j = t0 - (int64_t)m + c;
        const bool valid = j >= 0 && j + (int64_t)m < (int64_t)n;
        L[c] = valid ? (int8_t)lcp_digits(codes[c + 1u], codes[c + 1u + m], bits) : (int8_t)-1;
    }
    __syncthreads();
    const uint32_t k0 = threadIdx.x * 4u;
    uint32_t packed = 0u;
#pragma unroll
    for (uint32_t e = 0; e < 4u; ++e) {
        const uint32_t k = k0 + e;
        const int64_t i = t0 + k;
        uint32_t nn = 0u;
        if (i < (int64_t)n) {
            int D = -1;
            for (uint32_t q = 0; q <= m; ++q) { const int l = (int)L[k + q]; D = l > D ? l : D; }
            const int d0 = i == 0 ? 0 : lcp_digits(codes[k + m], codes[k + m + 1u], bits) + 1;
            const int d1 = D + 1 < bits ? D + 1 : bits;
            nn = d1 >= d0 ? (uint32_t)(d1 - d0 + 1) : 0u;
        }
        packed |= nn << (8u * e);
    }
    const int64_t i0 = t0 + k0;
    if (i0 + 3 < (int64_t)n) *reinterpret_cast<uint32_t*>(ncount + i0) = packed;  // (t0 and k0 are multiples of four)
    else
        for (uint32_t e = 0; e < 4u; ++e)
            if (i0 + e < (int64_t)n) ncount[i0 + e] = (uint8_t)(packed >> (8u * e));
}

// Only one item position in five or six starts a node, and what such a position costs is a chain of dependent loads (the galloping
// searches): a wave first gathers the positions of its stretch that do start nodes -- ballot + prefix count into its own piece of
// LDS -- and then works through that list with all lanes busy (3.0 -> 1 ms on the 75M-item list of BASELINE configs[4]).
constexpr uint32_t kNodeStretch = 512;  // item positions per wave

__global__ __launch_bounds__(256) void k_oct_nodes(const uint64_t* __restrict__ items, uint32_t n, int bits, const uint32_t* __restrict__ base,
                                                   vx_octree_node* __restrict__ nodes)
{
    __shared__ uint32_t live[4][kNodeStretch];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t p0 = ((uint64_t)blockIdx.x * 4u + wv) * kNodeStretch;
    if (p0 >= n) return;
    uint32_t nlive = 0;
#pragma unroll
    for (uint32_t k = 0; k < kNodeStretch / 64u; ++k) {
        const uint64_t p = p0 + k * 64u + lane;
        const bool act = p < n && base[p + 1u] != base[p];
        const unsigned long long m = __ballot(act);
        if (act) live[wv][nlive + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)p;
        nlive += (uint32_t)__popcll(m);
    }
    // (a wave's LDS writes are visible to its own later reads: no barrier)
    // The DEEPEST node of a position is a leaf (a child would start at the same position, one level deeper), no node starts inside
    // a leaf, and every node ends where another one starts: a leaf's run ends at the next position that starts nodes -- the next
    // entry of the list; for the stretch's last entry the first such position behind the stretch (n when there is none).
    uint32_t next_after = n;
    for (uint64_t q0 = p0 + kNodeStretch; q0 < n; q0 += 64u) {
        const uint64_t p = q0 + lane;
        const unsigned long long m = __ballot(p < n && base[p + 1u] != base[p]);
        if (m) { next_after = (uint32_t)(q0 + (uint64_t)(__ffsll((long long)m) - 1)); break; }
    }
  for (uint32_t e = lane; e < nlive; e += 64u) {
    const uint32_t i = live[wv][e];
    const uint32_t b0 = base[i], nn = base[i + 1u] - b0;
    const uint64_t code = items[i];
    const int d0 = i == 0u ? 0 : lcp_digits(items[i - 1u], code, bits) + 1;
    int64_t end_prev = (int64_t)n;
    for (uint32_t q = 0; q < nn; ++q) {
        const int d = d0 + (int)q;
        const int shift = 3 * (bits - d);  // 0 .. 63 (bits <= 21)
        const uint64_t pref = code >> shift;
        // end of the depth-d run that starts at i = the first position behind i whose prefix is LARGER (the items are sorted); it lies
        // inside the end of the shallower run.  Galloping search from i: runs are short as a rule.
        int64_t lo = (int64_t)i;   // prefix(item[lo]) == pref
        int64_t hi = end_prev;     // prefix(item[hi]) > pref, or hi == the shallower run's end
        if (q + 1u == nn) {
            hi = (int64_t)(e + 1u < nlive ? live[wv][e + 1u] : next_after);  // the leaf: no search
        } else {
#pragma nounroll
            for (int64_t step = 1; (int64_t)i + step < hi; step <<= 1) {
                const int64_t p = (int64_t)i + step;
                if ((items[p] >> shift) > pref) { hi = p; break; }
                lo = p;
            }
#pragma nounroll
            while (hi - lo > 1) {
                const int64_t mid = lo + ((hi - lo) >> 1);
                if ((items[mid] >> shift) > pref) hi = mid; else lo = mid;
            }
        }
        const uint32_t id = b0 + q;
        nodes[id].start = i;          // octTree.hpp:325-327
        nodes[id].count = (uint32_t)(hi - (int64_t)i);
        end_prev = hi;
        if (d > 0) {
            uint32_t parent;
            if (q > 0u) parent = id - 1u;
            else {
                // the node of depth d - 1 around i: its run starts at the smallest position whose first d - 1 digits are not SMALLER
                // than i's (galloping backwards from i)
                const int sp = shift + 3;
                const uint64_t pp = code >> sp;
                int64_t in = (int64_t)i;  // prefix(item[in]) == pp
                int64_t out = -1;         // prefix(item[out]) < pp, or -1
#pragma nounroll
                for (int64_t step = 1; step <= (int64_t)i; step <<= 1) {
                    const int64_t p = (int64_t)i - step;
                    if ((items[p] >> sp) < pp) { out = p; break; }
                    in = p;
                }
#pragma nounroll
                while (in - out > 1) {
                    const int64_t mid = out + ((in - out) >> 1);
                    if ((items[mid] >> sp) < pp) out = mid; else in = mid;
                }
                const int d0p = in == 0 ? 0 : lcp_digits(items[in - 1], items[in], bits) + 1;
                parent = base[in] + (uint32_t)((d - 1) - d0p);
            }
            nodes[parent].children[(uint32_t)(code >> shift) & 7u] = id;  // octTree.hpp:343,351
        }
    }
  }
}

// first position in [lo, hi) whose octant at `shift` is >= c
__device__ __forceinline__ uint32_t octant_lower_bound(const uint64_t* __restrict__ items, uint32_t lo, uint32_t hi, uint32_t shift, uint32_t c)
{
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint32_t)((items[mid] >> shift) & 7ull) < c) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// pass A: number of non-empty children of every node of the level (0 for leaves)
__global__ __launch_bounds__(256) void k_oct_count(const uint64_t* __restrict__ items, const uint32_t* __restrict__ start, const uint32_t* __restrict__ count,
                                                   uint32_t nnodes, uint32_t depth, uint32_t max_depth, uint32_t max_items, uint32_t* __restrict__ nchild)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nnodes) return;
    const uint32_t s = start[i], n = count[i];
    uint32_t k = 0;
    if (depth < max_depth && n > max_items) {  // octTree.hpp:329
        const uint32_t shift = 3u * (max_depth - 1u - depth);
        uint32_t prev = s;
        for (uint32_t c = 1; c <= 8; ++c) {
            const uint32_t b = c == 8 ? s + n : octant_lower_bound(items, prev, s + n, shift, c);
            k += b > prev;
            prev = b;
        }
    }
    nchild[i] = k;
}

// pass B: write the children into the next level and link them (breadth-first ids)
__global__ __launch_bounds__(256) void k_oct_emit(const uint64_t* __restrict__ items, uint32_t* __restrict__ start, uint32_t* __restrict__ count,
                                                  uint32_t* __restrict__ children /*8 per node*/, uint8_t* __restrict__ depth_of, uint32_t level_off,
                                                  uint32_t nnodes, uint32_t depth, uint32_t max_depth, uint32_t max_items,
                                                  const uint32_t* __restrict__ child_base, uint32_t next_off)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nnodes) return;
    const uint32_t id = level_off + i;
    const uint32_t s = start[id], n = count[id];
    depth_of[id] = (uint8_t)depth;
    uint32_t ch[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) ch[c] = 0xFFFFFFFFu;
    if (depth < max_depth && n > max_items) {
        const uint32_t shift = 3u * (max_depth - 1u - depth);
        uint32_t prev = s, k = child_base[i];
        for (uint32_t c = 1; c <= 8; ++c) {
            const uint32_t b = c == 8 ? s + n : octant_lower_bound(items, prev, s + n, shift, c);
            if (b > prev) {
                const uint32_t cid = next_off + k++;
                start[cid] = prev;
                count[cid] = b - prev;
                ch[c - 1] = cid;
            }
            prev = b;
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) children[(size_t)id * 8 + c] = ch[c];
}

__global__ __launch_bounds__(256) void k_oct_keys(const uint32_t* __restrict__ start, const uint8_t* __restrict__ depth_of, uint32_t n, uint64_t* __restrict__ keys,
                                                  uint32_t* __restrict__ ids)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((uint64_t)start[i] << 8) | depth_of[i];
    ids[i] = i;
}

__global__ __launch_bounds__(256) void k_oct_inverse(const uint32_t* __restrict__ order, uint32_t n, uint32_t* __restrict__ newidx)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r < n) newidx[order[r]] = r;
}

__global__ __launch_bounds__(256) void k_oct_finalize(const uint32_t* __restrict__ order, const uint32_t* __restrict__ newidx, const uint32_t* __restrict__ start,
                                                      const uint32_t* __restrict__ count, const uint32_t* __restrict__ children, uint32_t n,
                                                      vx_octree_node* __restrict__ out)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint32_t id = order[r];
    vx_octree_node nd;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t ch = children[(size_t)id * 8 + c];
        nd.children[c] = ch == 0xFFFFFFFFu ? 0xFFFFFFFFu : newidx[ch];
    }
    nd.start = start[id];
    nd.count = count[id];
    out[r] = nd;
}

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
};

hipError_t grow(Buf& b, size_t keep_bytes, size_t need_bytes, hipStream_t s)
{
    if (need_bytes <= b.cap) return hipSuccess;
    size_t ncap = b.cap ? b.cap : 4096;
    while (ncap < need_bytes) ncap *= 2;
    void* np = nullptr;
    hipError_t e = hipMalloc(&np, ncap);
    if (e != hipSuccess) return e;
    if (b.p && keep_bytes) {
        e = hipMemcpyAsync(np, b.p, keep_bytes, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (b.p) (void)hipFree(b.p);
    b.p = np;
    b.cap = ncap;
    return e;
}

}  // namespace

void launch_oct_depths(const uint64_t* items, uint32_t nitems, uint32_t bits, uint32_t max_items, uint8_t* ncount, hipStream_t s)
{
    if (!nitems) return;
    VX_KL(k_oct_depths, dim3((nitems + kDepthTile - 1) / kDepthTile), dim3(256), 0, s, items, nitems, (int)bits, max_items, ncount);
}

void launch_oct_nodes(const uint64_t* items, uint32_t nitems, uint32_t bits, const uint32_t* base, vx_octree_node* nodes, hipStream_t s)
{
    if (!nitems) return;
    VX_KL(k_oct_nodes, dim3((nitems + 4u * kNodeStretch - 1u) / (4u * kNodeStretch)), dim3(256), 0, s, items, nitems, (int)bits, base, nodes);
}

// Builds the node array on the device.  items: sorted Morton codes (device).  On success *nodes_out is a hipMalloc'ed array
// of *nnodes_out nodes (the caller frees it with hipFree).
hipError_t build_octree_nodes(const uint64_t* items, uint32_t nitems, uint32_t max_depth, uint64_t max_items_, vx_octree_node** nodes_out,
                              uint64_t* nnodes_out, hipStream_t s)
{
    const uint32_t max_items = max_items_ > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)max_items_;
    Buf start, count, children, depth_of, nchild, cbase, tmp;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() { for (Buf* b : {&start, &count, &children, &depth_of, &nchild, &cbase, &tmp}) if (b->p) (void)hipFree(b->p); };
#define OC(expr) do { e = (expr); if (e != hipSuccess) { cleanup(); return e; } } while (0)
    uint32_t total = 1, level_off = 0, level_n = 1;
    OC(grow(start, 0, 4096, s));
    OC(grow(count, 0, 4096, s));
    OC(grow(children, 0, 4096 * 8, s));
    OC(grow(depth_of, 0, 4096, s));
    const uint32_t root[2] = {0u, nitems};
    OC(hipMemcpyAsync(start.p, &root[0], 4, hipMemcpyHostToDevice, s));
    OC(hipMemcpyAsync(count.p, &root[1], 4, hipMemcpyHostToDevice, s));
    for (uint32_t depth = 0;; ++depth) {
        // pass A + scan -> size of the next level
        OC(grow(nchild, 0, ((size_t)level_n + 1) * 4, s));
        OC(grow(cbase, 0, ((size_t)level_n + 2) * 4, s));
        OC(grow(tmp, 0, scan_tmp_bytes(level_n) + 16, s));
        VX_KL(k_oct_count, dim3((level_n + 255) / 256), dim3(256), 0, s, items, (const uint32_t*)start.p + level_off, (const uint32_t*)count.p + level_off, level_n,
              depth, max_depth, max_items, (uint32_t*)nchild.p);
        unsigned long long* tot_dev = (unsigned long long*)((char*)tmp.p + scan_tmp_bytes(level_n));
        launch_scan_u32((const uint32_t*)nchild.p, (uint32_t*)cbase.p, level_n, false, tmp.p, tot_dev, s);
        unsigned long long next_n = 0;
        OC(hipMemcpyAsync(&next_n, tot_dev, 8, hipMemcpyDeviceToHost, s));
        OC(hipStreamSynchronize(s));
        if ((unsigned long long)total + next_n >= 0xFFFFFFFFull) { cleanup(); return hipErrorOutOfMemory; }
        const uint32_t next_off = total;
        const size_t need = (size_t)total + next_n;
        OC(grow(start, (size_t)total * 4, need * 4, s));
        OC(grow(count, (size_t)total * 4, need * 4, s));
        OC(grow(children, (size_t)total * 32, need * 32, s));
        OC(grow(depth_of, (size_t)total, need, s));
        VX_KL(k_oct_emit, dim3((level_n + 255) / 256), dim3(256), 0, s, items, (uint32_t*)start.p, (uint32_t*)count.p, (uint32_t*)children.p, (uint8_t*)depth_of.p,
              level_off, level_n, depth, max_depth, max_items, (const uint32_t*)cbase.p, next_off);
        if (next_n == 0) break;
        level_off = next_off;
        level_n = (uint32_t)next_n;
        total += level_n;
    }
    // pre-order numbering: rank of (start << 8 | depth)
    Buf keys, keys2, ids, order, newidx, sorttmp;
    auto cleanup2 = [&]() { for (Buf* b : {&keys, &keys2, &ids, &order, &newidx, &sorttmp}) if (b->p) (void)hipFree(b->p); };
#define OC2(expr) do { e = (expr); if (e != hipSuccess) { cleanup2(); cleanup(); return e; } } while (0)
    OC2(grow(keys, 0, (size_t)total * 8, s));
    OC2(grow(keys2, 0, (size_t)total * 8, s));
    OC2(grow(ids, 0, (size_t)total * 4, s));
    OC2(grow(order, 0, (size_t)total * 4, s));
    OC2(grow(newidx, 0, (size_t)total * 4, s));
    VX_KL(k_oct_keys, dim3((total + 255) / 256), dim3(256), 0, s, (const uint32_t*)start.p, (const uint8_t*)depth_of.p, total, (uint64_t*)keys.p, (uint32_t*)ids.p);
    size_t tb = 0;
    OC2(rocprim::radix_sort_pairs(nullptr, tb, (const uint64_t*)keys.p, (uint64_t*)keys2.p, (const uint32_t*)ids.p, (uint32_t*)order.p, (size_t)total, 0u, 40u, s));
    OC2(grow(sorttmp, 0, tb ? tb : 16, s));
    OC2(rocprim::radix_sort_pairs(sorttmp.p, tb, (const uint64_t*)keys.p, (uint64_t*)keys2.p, (const uint32_t*)ids.p, (uint32_t*)order.p, (size_t)total, 0u, 40u, s));
    VX_KL(k_oct_inverse, dim3((total + 255) / 256), dim3(256), 0, s, (const uint32_t*)order.p, total, (uint32_t*)newidx.p);
    vx_octree_node* out = nullptr;
    OC2(hipMalloc((void**)&out, (size_t)total * sizeof(vx_octree_node)));
    VX_KL(k_oct_finalize, dim3((total + 255) / 256), dim3(256), 0, s, (const uint32_t*)order.p, (const uint32_t*)newidx.p, (const uint32_t*)start.p,
          (const uint32_t*)count.p, (const uint32_t*)children.p, total, out);
    e = hipStreamSynchronize(s);
    cleanup2();
    cleanup();
    if (e != hipSuccess) { (void)hipFree(out); return e; }
    *nodes_out = out;
    *nnodes_out = total;
    return hipSuccess;
#undef OC
#undef OC2
}

}  // namespace vx
