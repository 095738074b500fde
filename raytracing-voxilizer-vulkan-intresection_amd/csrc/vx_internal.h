// vx_internal.h -- launch interface between the C ABI (vx_api.cpp) and the gfx950 kernels (vx_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vx_math.h"
#include "../../include/voxhip.h"

namespace vx {

// De-indexed triangle + its candidate voxel box, written once by k_tri_setup and read by every later pass.
// 48 B, 16-B aligned: three dwordx4 loads.
struct __attribute__((aligned(16))) TriRec {
    float v[9];     // v0.xyz v1.xyz v2.xyz                      loadPos, VoxelBuilder.hpp:356-362
    uint32_t xr;    // start | count << 16 along x               candidate range, VoxelBuilder.hpp:175-184
    uint32_t yr;
    uint32_t zr;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

// Cells per axis: the reference's grids are bounded by memory only, its Octree by 21 Morton bits per axis (octTree.hpp:583-585);
// candidate ranges are kept as 16 + 16 bits per axis in TriRec plus 5 + 5 high bits in a per-triangle extension word.
constexpr uint32_t kMaxDim = 1u << 21;
constexpr uint32_t kCoarse = 8;        // cells per brick edge, bricks per block edge (ray traversal structure)
constexpr uint32_t kCoarseShift = 3;
constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;          // elements per thread in the scan kernels
constexpr int kScanTile = kScanBlock * kScanItems;

struct Camera { float viewInv[16]; float projInv[16]; uint32_t width, height; };

// ---- optional per-kernel event timing (vx_prof.cpp) ----------------------------------------------------------
void prof_enable(bool on);
void prof_select(const char* name);  // nullptr/"" = all kernels
bool prof_enabled();
void prof_reset();
int prof_read(int slot, char* name, size_t cap, double* ms, uint64_t* n);
struct ProfScope {
    ProfScope(const char* name, hipStream_t s);
    ~ProfScope();
    const char* name_;
    hipStream_t s_;
    hipEvent_t a_;
};

// ---- launchers (all asynchronous on `s`) ------------------------------------------------------------------
// K1: bbox of all vertices with the reference's first-occurrence tie rule; out6 = min xyz, max xyz (device).
// state7: self-cleaning reduction state (initialise ONCE with bbox_state_init); out6 may point to pinned host memory;
// zero64 (optional): the per-build setVoxel-call counters (kCallCounters of them, one per 64-byte line), cleared by the kernel.
constexpr uint32_t kCallCounters = 64;  // k_voxelize's waves add to counter (wave index % 64): entry [8 * i] of the array
// dgrid (optional, device memory): origin = bbox min and dims = ceil((max - min) / vs) as the host will derive them, for kernels
// queued behind K1 before the host has read the bbox.
struct DevGrid {
    float org[3];
    uint32_t dim[3];
    uint32_t pad[2];
};
void bbox_state_init(unsigned long long state7[7]);
void launch_bbox(const float* verts, uint64_t nverts, unsigned long long* state7, float* out6, unsigned long long* zero64, hipStream_t s,
                 float vs = 1.0f, DevGrid* dgrid = nullptr);

// K2a: per-triangle record + number of row-segment work units; zlo/zhi clamp the candidate box to a z slab.
void launch_tri_setup(const float* verts, const int32_t* idx, uint64_t tri_begin, uint32_t ntri, const GridParams& g,
                      int sat_variant, uint32_t zlo, uint32_t zhi, TriRec* recs, uint32_t* units, hipStream_t s, const DevGrid* dgrid = nullptr,
                      void* clear = nullptr /*optional: a 16-byte aligned buffer the kernel zeroes beside its own work*/, uint64_t clear_bytes = 0,
                      uint64_t shard_wb = 0, uint64_t shard_we = 0 /*with dgrid: derive zlo / zhi on the device from the word shard (0, 0: whole grid)*/,
                      uint32_t* ext = nullptr /*optional, ntri words: bits 16..20 of the range values, for grids with an axis above 65535 cells*/,
                      uint32_t shard_rank = 0, uint32_t shard_world = 0 /*with dgrid and world > 1: the word shard vx_shard_words gives that rank*/);

// exclusive scan of n uint32 (or of their popcounts) into out[0..n] (out[n] = total, saturating check via *total64)
size_t scan_tmp_bytes(uint64_t n);
// tmp_is_zero: the caller guarantees tmp (scan_tmp_bytes(n)) is all zero; the scan leaves it all zero again.
// total_tag (bits 48..63 only): OR-ed into *total64 by the single-pass kernel, so that a host polling a pinned mailbox word can tell
// this scan's total from an older one; returns whether the tag was applied (false: the three-pass path, *total64 is the bare total).
bool launch_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, bool popcount_input, void* tmp,
                     unsigned long long* total64, hipStream_t s, bool tmp_is_zero = false, unsigned long long total_tag = 0,
                     uint32_t* sel1024 = nullptr /*optional (single-pass kernel only, values <= 1024): sel1024[c] = the element whose range holds c * 1024*/,
                     uint32_t gen = 0 /*generation mode (vx_kernels.hip): the number of this scan on `tmp`, 1, 2, 3 ... < 2^22; 0 = tickets + self-cleaning state*/,
                     uint32_t* group16 = nullptr /*optional (single-pass kernel only): group16[i] = out[16 i], n / 16 + 1 entries -- a dense array small
                                                   enough to stay in L2 for readers that gather (k_rank)*/);

void launch_scan_u8(const uint8_t* in, uint32_t* out, uint64_t n, void* tmp /*scan_tmp_bytes(n), all zero*/, unsigned long long* total64, hipStream_t s,
                    unsigned long long total_tag = 0, uint32_t gen = 0);

// K2: triangle/voxel overlap over all work units; ORs hits into `words` (only words in [wb,we)), optionally
// stores each unit's 32-bit hit mask (unit_mask) for the ordered emitters; adds the hit count to *set_calls.
// block_tri (optional): per 256-unit block the triangle of its first unit (launch_unit_blocks), enables LDS staging.
void launch_unit_blocks(const uint32_t* unit_base, uint32_t ntri, uint32_t total_units, uint32_t* block_tri, hipStream_t s, uint32_t cap_blocks = 0xFFFFFFFFu);
void launch_voxelize(const TriRec* recs, const uint32_t* unit_base, const uint32_t* block_tri, uint32_t ntri, const GridParams& g,
                     int sat_variant, uint32_t* words, uint64_t wb, uint64_t we, uint32_t* unit_mask, unsigned long long* set_calls,
                     hipStream_t s, const uint32_t* ext = nullptr /*k_tri_setup's extension words; null unless an axis has more than 65535 cells*/,
                     uint32_t* block_hits = nullptr /*with unit_mask: hits per block of 64 units, (U + 63) / 64 entries -- what launch_emit_units'
                                                      block_base is the exclusive scan of*/,
                     bool tiled = false /*`words` is the tiled build mask (tiled_mask_words(dim) words, dim[0] % 32 == 0, whole grid): launch_untile
                                          then writes the reference's bitmask*/);
uint64_t tiled_mask_words(const uint32_t dim[3]);
void launch_untile(const uint32_t* tiled, uint32_t* words, const uint32_t dim[3], hipStream_t s, uint64_t wb = 0, uint64_t we = ~0ull /*the words the build owns: the others are written as zero*/);

// K3: VoxelGridVec / Octree emitters: one output per set bit of unit_mask, in unit order (== reference order).  block_base[b] = position of
// the first hit of units [64 b, 64 b + 64): the exclusive scan of launch_voxelize's block_hits.
void launch_emit_units(const TriRec* recs, const uint32_t* unit_base, const uint32_t* block_tri, uint32_t ntri, const GridParams& g,
                       const uint32_t* unit_mask, const uint32_t* block_base, vx_aabb* aabbs, uint64_t* morton, hipStream_t s, uint64_t cap = ~0ull,
                       const uint32_t* ext = nullptr);

// Per-voxel material ids (the reference's commented-out addMatrialIfNeeded plumbing): pass 1 = last triangle per occupied voxel
// (last_tri[rank], zeroed by the caller; may be null for the Vec flavour) + tri_hit[t] = 1 for triangles that set a voxel (zeroed by
// the caller); pass 2 = triangle -> material value -> index of first use.
void launch_mat_last(const TriRec* recs, const uint32_t* unit_base, const uint32_t* block_tri, uint32_t ntri, const GridParams& g, const uint32_t* unit_mask,
                     const uint32_t* words, const uint32_t* word_prefix, uint32_t* last_tri, uint8_t* tri_hit, hipStream_t s, const uint32_t* ext = nullptr);
void launch_mat_ids(const uint32_t* last_tri, uint64_t n, const int32_t* tri_value, const int16_t* value_index, int16_t* out, hipStream_t s);
void launch_mat_ids_calls(const TriRec* recs, const uint32_t* unit_base, const uint32_t* block_tri, uint32_t ntri, const uint32_t* unit_mask,
                          const uint32_t* block_base, const int32_t* tri_value, const int16_t* value_index, int16_t* out, hipStream_t s);

// K4: bitmask -> ordered AABB list (word_prefix = exclusive scan of popcounts, nwords+1 entries)
void launch_emit_bool_aabbs(const uint32_t* words, const uint32_t* word_prefix, const GridParams& g, vx_aabb* out,
                            uint64_t capacity, hipStream_t s, const uint32_t* sel1024 = nullptr /*from the prefix scan: the word of every 1024th record*/);
// AABBs from sorted Morton items (Octree::getAabbs)
void launch_emit_morton_aabbs(const uint64_t* items, uint64_t n, const float root_min[3], float vs, vx_aabb* out, hipStream_t s);

// ---- K6: first hit per ray (vx_walk.hip) and the structure it walks --------------------------------------------------------
// Traversal structure = the analogue of the reference's BLAS build (hello_vulkan.cpp:737-760), three levels:
//   bricks3  the bitmask re-tiled brick-major (8^3 cells) in three orientations, one per possible major axis of a ray:
//            bricks3[ori][brick][slab along ori], ori 0 x / 1 y / 2 z; orientation 2 holds bit y*8+x per z slab, orientation 0
//            bit z*8+y per x slab, orientation 1 bit x*8+z per y slab
//   w1       one bit per brick (dims d1), x-fastest;   w2: one bit per 8^3 bricks (dims d2)
// (m1 given and bdim[0] % 64 == 0: the brick kernel writes the level-1 mip itself, leaves empty bricks unwritten and returns true)
bool launch_build_bricks3(const uint32_t* words, const uint32_t dim[3], const uint32_t bdim[3], unsigned long long* bricks3, uint32_t* m1, hipStream_t s,
                          const uint32_t* tiled = nullptr /*the voxelizer's tiled build mask (dim[0] % 32 == 0): the source instead of `words`, and `words`
                                                            is WRITTEN from it -- launch_untile's job done on the way*/);
void launch_brick_mip1(const unsigned long long* bricks_z /*orientation 2*/, uint64_t nbricks, uint32_t* m1, hipStream_t s);
void launch_build_mip2(const uint32_t* m1, const uint32_t d1[3], const uint32_t d2[3], uint32_t* m2, hipStream_t s);
struct TraceMips {
    const unsigned long long* bricks3;
    const uint32_t* w0;                // the reference-layout bitmask (primitive rank only)
    const uint32_t* w1;
    const uint32_t* w2;
    uint32_t d1[3];
    uint32_t d2[3];
};
struct TraceIO {
    const float* rays = nullptr;         // 6 f32 per ray, or null with cam
    const Camera* cam = nullptr;         // primary rays generated in-kernel (host copy)
    const Camera* cam_dev = nullptr;     // the same camera in device memory (filled in by the API layer)
    uint64_t nrays = 0;
    float tmin = 0.001f, tmax = 10000.0f;
    const float* tmax_per_ray = nullptr; // optional per-ray tMax
    bool any_hit = false;                // terminate on the first accepted hit (shadow query)
    float* t_out = nullptr;
    uint32_t* prim_out = nullptr;
    float* normal_out = nullptr;         // 3 f32 per ray: cube-face normal (zero for misses)
    uint8_t* shadowed_out = nullptr;     // 1 = some accepted hit
    vx_hit* hits = nullptr;
    unsigned long long* nhits = nullptr;
};
// voxel indices of the hits travel from the walk to the rank pass as 32-bit words when they fit (all ones = miss)
inline bool trace_idx32(const GridParams& g) { return g.nvox < 0xFFFFFFFFull; }
inline size_t trace_idx_bytes(const GridParams& g, uint64_t nrays) { return (size_t)nrays * (trace_idx32(g) ? 4 : 8) + 8; }
// The ray kernel's work queue, for whoever wants to start something when it runs dry (hipStreamWaitValue64 on another stream): the device
// word the kernel's waves draw their chunks of rays from and the value it has reached when the last chunk is out -- from then on the launch
// only drains (0: the static first chunks cover the batch).  Batches of more than 2^31 rays: the last launch's.
struct WalkQueue {
    unsigned long long* counter = nullptr;
    unsigned long long dry_at = 0;
};
void launch_queue_gate(const WalkQueue& q, unsigned long long at_least, unsigned timeout_us, hipStream_t s);  // vx_trace.hip
void launch_trace(const GridParams& g, const TraceMips& mips, const uint32_t* word_prefix, const TraceIO& io, unsigned long long* counters /*2 device words, zero before the first trace; they alternate*/, int* phase /*host*/,
                  void* idx_tmp /*trace_idx_bytes when ranks / normals / the hit list are wanted*/, hipStream_t s,
                  const uint32_t* prefix16 = nullptr /*optional: launch_scan_u32's group16 of word_prefix*/, WalkQueue* queue = nullptr);

// single-voxel helpers
void launch_set_bit(uint32_t* words, uint64_t idx, hipStream_t s);

// device radix sort of uint64 keys (octree items; vx_sort.hip); tmp sized by sort_tmp_bytes.  The two key buffers ping-pong:
// returns 0 when the sorted keys end up in keys_a, 1 for keys_b (the other buffer is scratch afterwards).
size_t sort_tmp_bytes(uint64_t n);
int launch_sort_u64(uint64_t* keys_a /*in*/, uint64_t* keys_b, uint64_t n, int bits, void* tmp, size_t tmp_bytes, hipStream_t s);

// Octree node array, direct form (max_items <= kOctDirectMaxItems): per item position the number of nodes that START there
// (launch_oct_depths, bytes), an exclusive scan of those = the pre-order index of the first of them, then start / count / child links
// (launch_oct_nodes) into a node array the caller has filled with 0xFF bytes.
constexpr uint64_t kOctDirectMaxItems = 64;
void launch_oct_depths(const uint64_t* items, uint32_t nitems, uint32_t bits, uint32_t max_items, uint8_t* ncount, hipStream_t s);
void launch_oct_nodes(const uint64_t* items, uint32_t nitems, uint32_t bits, const uint32_t* base, vx_octree_node* nodes, hipStream_t s);

// Octree node array (pre-order, octTree.hpp:319-358) built on the device from the sorted items, level by level (any max_items);
// *nodes_out is hipMalloc'ed.
hipError_t build_octree_nodes(const uint64_t* items, uint32_t nitems, uint32_t max_depth, uint64_t max_items, vx_octree_node** nodes_out,
                              uint64_t* nnodes_out, hipStream_t s);

}  // namespace vx
