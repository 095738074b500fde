// vx_api.cpp -- the C ABI declared in include/voxhip.h: handles, device-memory pool, launch sequencing.
// All compute happens in the kernels of vx_kernels.hip; there is no CPU compute path in this library.
#include "vx_internal.h"

#include <cmath>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#pragma clang fp contract(off)

namespace vx {
int load_obj(const char* path, std::vector<float>& verts, std::vector<int32_t>& tris, std::vector<int32_t>& tri_mat, std::vector<vx_material>& mats,
             std::string& msg);
}

namespace {

thread_local std::string g_err;
thread_local int g_device = 0;

vx_status fail(vx_status s, const std::string& m)
{
    g_err = m;
    return s;
}

#define VX_HIP(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) return fail(VX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define VX_TRY(expr)                 \
    do {                             \
        vx_status s__ = (expr);      \
        if (s__ != VX_OK) return s__; \
    } while (0)

// ---- pooled device memory: hipMalloc/hipFree are slow and synchronising, steady-state loops must not call them ----
// The pool is STREAM-ORDERED: a block goes back on the free list together with the stream its owner queued work on and an
// event recorded on that stream at the moment of the release.  It is handed out again at once to a request from the SAME
// stream (work queued later on that stream runs after the old owner's kernels), and to any other stream only once the
// event has completed -- so a block that kernels in flight still read or write is never given to a different handle
// (distinct handles on distinct threads with per-handle streams are allowed by voxhip.h).
constexpr int kMaxDev = 16;
struct FreeBlock {
    void* p;
    hipStream_t stream;
    hipEvent_t ev;  // null: nothing was in flight (never used, or released after a synchronize)
};
struct Pool {
    std::mutex mu;
    std::multimap<size_t, FreeBlock> free_blocks;
    std::unordered_map<void*, size_t> live;
    std::vector<hipEvent_t> spare_events;
};
Pool g_pool[kMaxDev];

size_t round_size(size_t b)
{
    if (b < 256) b = 256;
    if (b >= (1u << 20)) return (b + (2u << 20) - 1) / (2u << 20) * (2u << 20);
    size_t p = 256;
    while (p < b) p <<= 1;
    return p;
}

hipError_t pool_alloc(int dev, size_t bytes, void** out, hipStream_t stream)
{
    const size_t sz = round_size(bytes);
    Pool& P = g_pool[dev];
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (auto it = P.free_blocks.lower_bound(sz); it != P.free_blocks.end() && it->first <= sz * 2; ++it) {
            FreeBlock& fb = it->second;
            if (fb.ev && fb.stream != stream && hipEventQuery(fb.ev) != hipSuccess) continue;  // still in flight on another stream
            if (fb.ev) P.spare_events.push_back(fb.ev);
            *out = fb.p;
            P.live[*out] = it->first;
            P.free_blocks.erase(it);
            return hipSuccess;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
        // drop the cache and retry once
        vx_release_cached_memory();
        e = hipMalloc(&p, sz);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(P.mu);
    P.live[p] = sz;
    *out = p;
    return hipSuccess;
}

// `in_flight`: work that touches the block may still be queued on `stream` (false: the caller has synchronized)
void pool_free(int dev, void* p, hipStream_t stream, bool in_flight)
{
    if (!p) return;
    Pool& P = g_pool[dev];
    hipEvent_t ev = nullptr;
    if (in_flight) {
        {
            std::lock_guard<std::mutex> lk(P.mu);
            if (!P.spare_events.empty()) { ev = P.spare_events.back(); P.spare_events.pop_back(); }
        }
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
        if (!ev || hipEventRecord(ev, stream) != hipSuccess) {
            // no event to order the reuse by: fall back to waiting for the stream
            (void)hipStreamSynchronize(stream);
            if (ev) { std::lock_guard<std::mutex> lk(P.mu); P.spare_events.push_back(ev); }
            ev = nullptr;
        }
    }
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) { if (ev) P.spare_events.push_back(ev); return; }
    P.free_blocks.emplace(it->second, FreeBlock{p, stream, ev});
    P.live.erase(it);
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int dev = 0;
    hipStream_t stream = nullptr;  // the stream the owner queues this buffer's work on (set by the owning handle)
    bool fresh = false;  // set when ensure() handed out a new block (contents undefined); the owner clears it
    uint32_t scan_gen = 0;  // scan scratch only: the number of the last scan that used this block (generation mode of the single-pass scan)
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap && p) return hipSuccess;
        release();
        hipError_t e = pool_alloc(dev, bytes, &p, stream);
        if (e == hipSuccess) { cap = round_size(bytes); fresh = true; }
        return e;
    }
    void release(bool in_flight = true)
    {
        if (p) pool_free(dev, p, stream, in_flight);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; }
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

vx_status need_device(int dev)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(VX_ERR_NO_DEVICE, "no HIP device available: libvoxhip has no CPU path");
    if (dev < 0 || dev >= n || dev >= kMaxDev) return fail(VX_ERR_INVALID_ARG, "device index out of range");
    return VX_OK;
}

// small per-handle scratch in device memory
struct Small {
    unsigned long long bbox_state[8];  // K1's self-cleaning reduction state (initialised once, see ensure_small)
    vx::DevGrid dgrid;                 // origin + dims as K1 derives them, for kernels queued before the host has seen the bbox
    unsigned long long set_calls[vx::kCallCounters * 8];  // 64 counters on lines of their own (k_voxelize), summed by sync_counts
    unsigned long long nhits;
    unsigned long long trace_counters[4];
};

// The few values the HOST waits for (bbox -> grid dims, unit / hit / occupied counts -> buffer sizes).  Kernels write them
// straight into pinned host memory; the host reads them after a stream synchronize.  No device-to-host copy kernel, no
// staging through pageable memory.
struct Mail {
    float bbox[8];
    unsigned long long units, hits, occupied, pad;
};

// Totals may arrive tagged with the build's sequence number in bits 48..63 (see launch_scan_u32): the host then polls the word
// instead of draining the stream -- the totals are written by scans that finish long before the build's last kernel.
constexpr unsigned long long kMailValue = (1ull << 48) - 1ull;
inline bool mail_wait(const volatile unsigned long long* a, const volatile unsigned long long* b /*optional*/, unsigned long long tag, double timeout_ms)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        if ((*a & ~kMailValue) == tag && (!b || (*b & ~kMailValue) == tag)) return true;
        if ((spin & 255u) == 255u && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) return false;
        __builtin_ia32_pause();
    }
}

hipError_t mail_alloc(Mail** out)
{
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, sizeof(Mail), hipHostMallocCoherent);
    if (e != hipSuccess) return e;
    std::memset(p, 0, sizeof(Mail));
    *out = reinterpret_cast<Mail*>(p);
    return hipSuccess;
}

// (re)allocated Small: the bbox reduction state needs its initial values once; everything else starts at zero
hipError_t ensure_small(DevBuf& small)
{
    hipError_t e = small.ensure(sizeof(Small));
    if (e != hipSuccess || !small.fresh) return e;
    Small h;
    std::memset(&h, 0, sizeof(h));
    vx::bbox_state_init(h.bbox_state);
    e = hipMemcpy(small.p, &h, sizeof(h), hipMemcpyHostToDevice);
    if (e == hipSuccess) small.fresh = false;
    return e;
}

// scan scratch with the "all zero between scans" contract of launch_scan_u32(..., tmp_is_zero = true)
hipError_t ensure_scan_tmp(DevBuf& tmp, size_t bytes, hipStream_t s)
{
    hipError_t e = tmp.ensure(bytes);
    if (e != hipSuccess || !tmp.fresh) return e;
    e = hipMemsetAsync(tmp.p, 0, tmp.cap, s);
    if (e == hipSuccess) { tmp.fresh = false; tmp.scan_gen = 0; }
    return e;
}

// the number of the next scan on this scratch block (generation mode: state words of older scans read as "not there yet")
uint32_t next_scan_gen(DevBuf& tmp, hipStream_t s)
{
    static const bool off = getenv("VOXHIP_SCAN_GEN") && atoi(getenv("VOXHIP_SCAN_GEN")) == 0;  // 0: tickets + self-cleaning state (A/B, tests)
    if (off) return 0u;
    if (++tmp.scan_gen >= (1u << 22)) {
        (void)hipMemsetAsync(tmp.p, 0, tmp.cap, s);
        tmp.scan_gen = 1u;
    }
    return tmp.scan_gen;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
struct vx_mesh {
    int device = 0;
    std::vector<float> hv;
    std::vector<int32_t> hi;
    std::vector<int32_t> tri_mat;        // material id per triangle (-1 = none); empty: the mesh has no materials
    std::vector<vx_material> materials;  // m_materials of the reference builder (VoxelBuilder.hpp:69)
    // material VALUES for the voxelizer: the mesh's records de-duplicated by MaterialObj::operator== (value 0 is always the default
    // MaterialObj{} that faces without usemtl carry, VoxelBuilder.hpp:383) and the value id of every triangle
    std::vector<vx_material> values;
    std::vector<int32_t> tri_value;
    bool values_ready = false;
    DevBuf btv;  // tri_value on the device
    size_t nv = 0, nt = 0;
    const float* dv = nullptr;
    const int32_t* di = nullptr;
    DevBuf bv, bi;
    bool borrowed = false;
    bool uploaded = false;
};

namespace {
vx_material default_material()  // MaterialObj{} (common/obj_loader.h:32-43)
{
    vx_material m;
    std::memset(&m, 0, sizeof(m));
    m.ambient[0] = m.ambient[1] = m.ambient[2] = 0.1f;
    m.diffuse[0] = m.diffuse[1] = 1.0f;
    m.specular[0] = m.specular[1] = m.specular[2] = 1.0f;
    m.emission[2] = 0.10f;
    m.shininess = 0.f;
    m.ior = 1.0f;
    m.dissolve = 1.f;
    m.illum = 0;
    m.texture_id = -1;
    return m;
}
bool same_material(const vx_material& a, const vx_material& b)  // MaterialObj::operator== (obj_loader.h:45-51): ior and dissolve are not compared
{
    for (int k = 0; k < 3; ++k)
        if (a.ambient[k] != b.ambient[k] || a.diffuse[k] != b.diffuse[k] || a.specular[k] != b.specular[k] || a.transmittance[k] != b.transmittance[k] ||
            a.emission[k] != b.emission[k])
            return false;
    return a.shininess == b.shininess && a.illum == b.illum && a.texture_id == b.texture_id;
}
}  // namespace

struct vx_grid {
    int device = 0;
    hipStream_t stream = nullptr;
    int kind = VX_GRID_BOOL;
    vx::GridParams g{};
    float bbmin[3] = {0, 0, 0}, bbmax[3] = {0, 0, 0}, bbc[3] = {0, 0, 0};
    uint64_t triangles = 0;
    uint32_t cdim[3] = {0, 0, 0}, c2dim[3] = {0, 0, 0};
    DevBuf words, twords /*tiled build mask (launch_voxelize)*/, cwords, c2words, bricks, idxtmp, ttmp, camera, wprefix, wsel /*word of every 1024th occupied voxel (prefix scan)*/, wp16 /*every 16th entry of wprefix, dense (prefix scan)*/, recs, ext /*high bits of the candidate ranges*/, units, ubase, btri, umask, bhits /*hits per block of 64 units*/, hbase /*their exclusive scan*/, scantmp, small, vec, matids, mattmp;
    std::vector<vx_material> materials;  // m_materials: distinct values in first-use order (VX_VOXELIZE_MATERIALS builds only)
    uint64_t mat_count = 0;              // entries of matids
    bool has_materials = false;
    // a VX_VOXELIZE_MATERIALS build in two halves: what the build itself knows (per value the first triangle of this shard that uses
    // it; the last triangle per voxel resp. the unit masks stay in mattmp / umask / hbase) and the finish (finish_materials)
    bool mat_pending = false;
    std::vector<long long> mat_first_use;
    std::vector<vx_material> mat_values;   // the mesh's material values at build time
    const int32_t* mat_dtv = nullptr;      // per-triangle value ids of this build's triangle range (device, owned by the mesh)
    uint32_t mat_ntri = 0;
    uint64_t mat_nids = 0;
    bool mat_gathered = false;             // multi-GPU build: the ids of ALL shards, gathered in shard order, live in mattmp
    uint64_t mat_gather_count = 0;
    bool coarse_valid = false, prefix_valid = false /*word_prefix queued or done*/, occupied_known = false, counts_valid = true;
    bool last_tiled = false;  // the previous build on this handle went through the tiled build mask (twords)
    uint64_t occupied = 0, set_calls = 0, host_set_calls = 0;
    uint64_t vec_count = 0;
    uint32_t mail_seq = 0;  // sequence tag of the totals the current build writes to the mailbox
    bool sel_valid = false;  // wsel belongs to the current word prefix
    unsigned long long occ_tag = 0;  // tag of the word-prefix scan whose total (the occupied count) is in flight; 0: untagged
    int trace_phase = 0;  // which of Small::trace_counters[0..1] the next ray launch draws its work from (the launch clears the other)
    // VX_GRID_VEC: the caller's own list buffer (vx_grid_bind_aabbs_device); builds emit straight into it when it is large enough
    vx_aabb* bound = nullptr;
    uint64_t bound_cap = 0;
    bool vec_in_bound = false;  // the current list lives in `bound`, not in `vec`
    vx_aabb* vec_ptr() const { return vec_in_bound ? bound : reinterpret_cast<vx_aabb*>(vec.p); }
    // VX_VOXELIZE_LIST_ASYNC (VX_GRID_VEC): the ordered emission of the list is not queued by the build.  Nothing that follows a build on the
    // handle's stream -- traversal structure, word prefix, a ray batch -- reads the list; the ray kernel keeps a set of persistent workgroups
    // on every CU and ends with a long drain in which most of the machine idles, and the emission (store-bound, 40 us on the bench scene) fits
    // into that: the next ray batch queues it on a low-priority SIDE stream of the handle, behind an event recorded on the main stream in front
    // of the ray kernel, so that its workgroups fill the slots the ray kernel's leave.  Whatever reads the list or overwrites what the emission
    // reads (host / device copies of the list, re-binding, setVoxel, the next build, free) goes through list_resolve() first.
    hipStream_t side = nullptr;
    hipEvent_t ev_ready = nullptr, ev_list = nullptr;
    bool list_deferred = false;  // the emission has not been queued yet (its arguments: ld)
    bool list_pending = false;   // it has been queued on `side`; the main stream has not waited for ev_list yet
    struct { uint32_t ntri = 0; bool ext = false; vx_aabb* tgt = nullptr; uint64_t cap = 0; bool from_mask = false; } ld;  // from_mask: vx_grid_aabbs_device_async (K4)
    hipError_t side_init()
    {
        if (side) return hipSuccess;
        int lo = 0, hi = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);  // (lo: numerically greatest = lowest priority)
        static const bool low_prio = !(getenv("VOXHIP_LIST_SIDE_PRIO") && atoi(getenv("VOXHIP_LIST_SIDE_PRIO")) == 0);
        static const bool light_ev = !(getenv("VOXHIP_LIST_LIGHT_EVENTS") && atoi(getenv("VOXHIP_LIST_LIGHT_EVENTS")) == 0);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&side, hipStreamNonBlocking, low_prio ? lo : hi);
        // (both events order streams of ONE device: no system-scope fence -- its cache write-back would stand in front of the ray kernel)
        const unsigned evf = hipEventDisableTiming | (light_ev ? hipEventDisableSystemFence : 0u);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_ready, evf);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_list, evf);
        return e;
    }
    void list_emit(hipStream_t st)
    {
        if (ld.from_mask) {  // Bool / AABBstruct: the ascending list from the bitmask and its word prefix (both complete on the main stream)
            vx::launch_emit_bool_aabbs(words.as<uint32_t>(), wprefix.as<uint32_t>(), g, ld.tgt, ld.cap, st, sel_valid ? wsel.as<uint32_t>() : nullptr);
            return;
        }
        vx::launch_emit_units(recs.as<vx::TriRec>(), ubase.as<uint32_t>(), btri.as<uint32_t>(), ld.ntri, g, umask.as<uint32_t>(), hbase.as<uint32_t>(), ld.tgt, nullptr, st,
                              ld.cap, ld.ext ? ext.as<uint32_t>() : nullptr);
    }
    // first half, BEFORE the caller queues the work the emission is to run beside; second half after it
    hipError_t list_side_begin()
    {
        hipError_t e = side_init();
        if (e == hipSuccess) e = hipEventRecord(ev_ready, stream);
        return e;
    }
    // `counter`, `dry_at`: the ray kernel's work queue (vx::WalkQueue), for the gate in front of the emission.  Holding the emission until the
    // queue is DRY (VOXHIP_LIST_WAIT=2) was measured and is not the default: the drain frees registers wave by wave but LDS only workgroup
    // by workgroup, the emission then needs the whole drain (155 us) and ends about when the ray kernel does -- 0.504-0.510 ms per step.
    hipError_t list_side_launch(unsigned long long* counter, unsigned long long dry_at)
    {
        hipError_t e = hipStreamWaitEvent(side, ev_ready, 0);
        if (e != hipSuccess) return e;
        // VOXHIP_LIST_WAIT: 0 no hold; 1 (default) until the first wave of the ray kernel has come back to the queue for more rays -- the
        // kernel's persistent workgroups are all placed by then, the emission cannot take their slots first (without the hold the step
        // varies 0.497-0.535 ms from run to run, with it 0.489-0.494); 2 until the queue is dry (measured: slower).  The hold is a
        // one-wave gate kernel with a time bound (vx_trace.hip: a stream-level wait on the counter hangs under serialising profilers).
        static const int wait_mode = getenv("VOXHIP_LIST_WAIT") ? atoi(getenv("VOXHIP_LIST_WAIT")) : 1;
        if (wait_mode && counter && dry_at) {  // (dry_at == 0: the static first chunks cover the batch, the counter never moves)
            vx::WalkQueue q;
            q.counter = counter;
            q.dry_at = dry_at;
            vx::launch_queue_gate(q, wait_mode == 2 ? dry_at : 1ull, /*timeout_us=*/wait_mode == 2 ? 2000u : 300u, side);
        }
        list_emit(side);
        e = hipEventRecord(ev_list, side);
        list_deferred = false;
        list_pending = true;
        return e;
    }
    // work queued on the main stream after this call sees the complete list (and may overwrite what the emission read)
    hipError_t list_resolve(bool drop_unqueued = false)
    {
        if (list_deferred) {
            list_deferred = false;
            // nobody asked for rays in between: the emission runs where it always did (a caller's buffer is always filled; the grid's own
            // Vec list may be dropped when it is about to be replaced)
            if (!drop_unqueued || ld.from_mask) list_emit(stream);
        }
        if (list_pending) {
            list_pending = false;
            return hipStreamWaitEvent(stream, ev_list, 0);
        }
        return hipSuccess;
    }
    void side_release()
    {
        if (!side) return;
        (void)hipStreamSynchronize(side);
        (void)hipStreamDestroy(side);
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_list) (void)hipEventDestroy(ev_list);
        side = nullptr;
        ev_ready = ev_list = nullptr;
        list_deferred = list_pending = false;
    }
    Mail* mail = nullptr;
    void set_dev(int d)
    {
        device = d;
        for (DevBuf* b : {&words, &twords, &cwords, &c2words, &bricks, &idxtmp, &ttmp, &camera, &wprefix, &wsel, &wp16, &recs, &ext, &units, &ubase, &btri, &umask, &bhits, &hbase, &scantmp, &small, &vec, &matids, &mattmp}) b->dev = d;
    }
    // the stream this handle queues work on; the pool orders the reuse of released blocks by it
    void set_stream(hipStream_t st)
    {
        if (st != stream && words.p) {  // work queued on the old stream (and beside it) must not outlive the switch
            (void)list_resolve();
            (void)hipStreamSynchronize(stream);
        }
        stream = st;
        for (DevBuf* b : {&words, &twords, &cwords, &c2words, &bricks, &idxtmp, &ttmp, &camera, &wprefix, &wsel, &wp16, &recs, &ext, &units, &ubase, &btri, &umask, &bhits, &hbase, &scantmp, &small, &vec, &matids, &mattmp}) b->stream = st;
    }
    void release_all()
    {
        for (DevBuf* b : {&words, &twords, &cwords, &c2words, &bricks, &idxtmp, &ttmp, &camera, &wprefix, &wsel, &wp16, &recs, &ext, &units, &ubase, &btri, &umask, &bhits, &hbase, &scantmp, &small, &vec, &matids, &mattmp}) b->release();
        if (mail) (void)hipHostFree(mail);
        mail = nullptr;
    }
};

struct vx_octree {
    int device = 0;
    hipStream_t stream = nullptr;
    float vs = 0.f;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    uint64_t dim[3] = {0, 0, 0};
    uint32_t bits = 0;
    uint64_t max_items = 16;
    uint64_t nitems = 0;
    DevBuf items;  // sorted Morton codes
    vx_octree_node* dnodes = nullptr;  // pre-order node array (device): in `nodebuf` (pooled) or hipMalloc'ed (level-by-level build)
    DevBuf nodebuf;
    uint64_t nnodes = 0;
    void free_nodes(bool in_flight)
    {
        if (nodebuf.p) nodebuf.release(in_flight);
        else if (dnodes) (void)hipFree(dnodes);
        dnodes = nullptr;
    }
};

namespace {

vx_status mesh_to_device(vx_mesh* m)
{
    if (m->uploaded || m->borrowed) return VX_OK;
    VX_TRY(need_device(m->device));
    m->bv.dev = m->bi.dev = m->device;
    VX_HIP(m->bv.ensure(m->nv * 12 + 16));
    VX_HIP(m->bi.ensure(m->nt * 12 + 16));
    if (m->nv) VX_HIP(hipMemcpy(m->bv.p, m->hv.data(), m->nv * 12, hipMemcpyHostToDevice));
    if (m->nt) VX_HIP(hipMemcpy(m->bi.p, m->hi.data(), m->nt * 12, hipMemcpyHostToDevice));
    m->dv = m->bv.as<float>();
    m->di = m->bi.as<int32_t>();
    m->uploaded = true;
    return VX_OK;
}

// material values of a mesh (host) and their per-triangle ids on the device
vx_status mesh_material_values(vx_mesh* m)
{
    if (!m->values_ready) {
        if (m->borrowed && !m->tri_mat.empty() && m->tri_mat.size() != m->nt) return fail(VX_ERR_INVALID_ARG, "material ids do not match the triangle count");
        m->values.clear();
        m->values.push_back(default_material());
        std::vector<int32_t> rec_value(m->materials.size(), 0);
        for (size_t i = 0; i < m->materials.size(); ++i) {
            // VoxelBuilder copies the record into a fresh MaterialObj (VoxelBuilder.hpp:383-394): textureID stays -1
            vx_material v = m->materials[i];
            v.texture_id = -1;
            int32_t id = -1;
            for (size_t k = 0; k < m->values.size(); ++k)
                if (same_material(m->values[k], v)) { id = (int32_t)k; break; }
            if (id < 0) { m->values.push_back(v); id = (int32_t)m->values.size() - 1; }
            rec_value[i] = id;
        }
        if (m->values.size() > 32767) return fail(VX_ERR_CAPACITY, "more than 32767 distinct materials: per-voxel ids are int16 (voxelgrid.hpp:29)");
        m->tri_value.assign(m->nt, 0);
        if (!m->tri_mat.empty())
            for (size_t t = 0; t < m->nt; ++t) {
                const int32_t id = m->tri_mat[t];
                m->tri_value[t] = (id >= 0 && (size_t)id < rec_value.size()) ? rec_value[(size_t)id] : 0;  // VoxelBuilder.hpp:384: out-of-range ids keep the default
            }
        m->values_ready = true;
        m->btv.release();
    }
    if (!m->btv.p && m->nt) {
        m->btv.dev = m->device;
        VX_HIP(m->btv.ensure(m->nt * 4 + 16));
        VX_HIP(hipMemcpy(m->btv.p, m->tri_value.data(), m->nt * 4, hipMemcpyHostToDevice));
    }
    return VX_OK;
}

// bbox + dims: computeBboxFromAttrib (VoxelBuilder.hpp:198-224) on the device, dims on the host (:347-349)
struct Extent {
    float mn[3], mx[3], ctr[3];
    uint64_t dim[3];
};

// morton_error: the caller is the Octree, whose limit of 2^21 cells per axis carries the reference's own message (octTree.hpp:583-585)
vx_status extent_from_bbox(const float* bb, size_t nv, float vs, Extent* e, bool morton_error = false)
{
    for (int a = 0; a < 3; ++a) {
        e->mn[a] = bb[a];
        e->mx[a] = bb[3 + a];
        e->ctr[a] = (bb[a] + bb[3 + a]) * 0.5f;  // VoxelBuilder.hpp:221
        if (nv == 0) { e->dim[a] = 0; continue; }
        const float q = std::ceil((e->mx[a] - e->mn[a]) / vs);  // :347-349
        if (!(q >= 0.0f) || q > (float)vx::kMaxDim) {
            if (morton_error && q > (float)vx::kMaxDim) return fail(VX_ERR_MORTON_BITS, "We support up to 21 bits per axis (max 2^21 voxels per dimension)!");
            return fail(VX_ERR_CAPACITY, "grid dimension outside [0, 2^21]: voxel size too small for this mesh");
        }
        e->dim[a] = (uint64_t)q;
    }
    return VX_OK;
}

vx_status compute_extent(const vx_mesh* m, float vs, Small* dsmall, Mail* mail, hipStream_t s, Extent* e, bool morton_error = false)
{
    // one kernel: reduction, result into the host mailbox, state restored, per-build setVoxel counter cleared
    vx::launch_bbox(m->dv, m->nv, dsmall->bbox_state, mail->bbox, dsmall->set_calls, s);
    VX_HIP(hipStreamSynchronize(s));
    return extent_from_bbox(mail->bbox, m->nv, vs, e, morton_error);
}

vx_status check_voxel_size(float vs)
{
    if (!(vs > 0.0f) || !std::isfinite(vs)) return fail(VX_ERR_INVALID_ARG, "voxel size must be a finite positive float");
    return VX_OK;
}

void fill_params(vx::GridParams& g, const float org[3], float vs, const uint64_t dim[3])
{
    for (int a = 0; a < 3; ++a) { g.org[a] = org[a]; g.dim[a] = (uint32_t)dim[a]; }
    g.vs = vs;
    g.half = vs * 0.5f;
    g.nvox = dim[0] * dim[1] * dim[2];
    g.nwords = (g.nvox + 31) / 32;
}

constexpr uint64_t kMaxVoxels = 1ull << 37;  // 16 GiB of bitmask

// The shared front half of buildVoxelGrid for grids and the octree: records, unit counts, unit bases.  In two parts so that
// the caller can queue work that does not depend on the unit count (clearing the bitmask) before the host waits for it.
vx_status setup_launch(const vx_mesh* m, const vx::GridParams& g, int sat, uint64_t tb, uint32_t ntri, uint32_t zlo, uint32_t zhi, DevBuf& recs,
                       DevBuf& units, DevBuf& ubase, DevBuf& scantmp, Mail* mail, hipStream_t s, DevBuf& ext, const vx::DevGrid* dgrid = nullptr,
                       unsigned long long mail_tag = 0, bool* tagged = nullptr, void* clear = nullptr, uint64_t clear_bytes = 0,
                       uint64_t shard_wb = 0, uint64_t shard_we = 0, uint32_t shard_rank = 0, uint32_t shard_world = 0)
{
    VX_HIP(recs.ensure((size_t)ntri * sizeof(vx::TriRec) + 64));
    VX_HIP(ext.ensure(((size_t)ntri + 1) * 4));  // high bits of the candidate ranges (read only when an axis has more than 65535 cells)
    VX_HIP(units.ensure(((size_t)ntri + 1) * 4));
    VX_HIP(ubase.ensure(((size_t)ntri + 2) * 4));
    VX_HIP(ensure_scan_tmp(scantmp, vx::scan_tmp_bytes(ntri), s));
    vx::launch_tri_setup(m->dv, m->di, tb, ntri, g, sat, zlo, zhi, recs.as<vx::TriRec>(), units.as<uint32_t>(), s, dgrid, clear, clear_bytes, shard_wb, shard_we,
                         ext.as<uint32_t>(), shard_rank, shard_world);
    const bool tg = vx::launch_scan_u32(units.as<uint32_t>(), ubase.as<uint32_t>(), ntri, false, scantmp.p, &mail->units, s, true, mail_tag, nullptr,
                                        next_scan_gen(scantmp, s));
    if (tagged) *tagged = tg && mail_tag != 0;
    return VX_OK;
}

// btri_entries: entries of the block table a launch queued before the host knew the total has already filled (0: none)
vx_status setup_finish(uint32_t ntri, DevBuf& ubase, DevBuf& btri, Mail* mail, hipStream_t s, uint64_t* total_units, bool stream_is_drained = false,
                       uint64_t btri_entries = 0)
{
    if (!stream_is_drained) VX_HIP(hipStreamSynchronize(s));
    const unsigned long long tot = mail->units & kMailValue;
    if (tot >= 0xFFFFFFFFull) return fail(VX_ERR_CAPACITY, "more than 2^32 candidate row segments: shard the mesh or the grid");
    *total_units = tot;
    if (tot && btri_entries < tot / 64 + 2) {
        VX_HIP(btri.ensure((size_t)(tot / 64 + 2) * 4));
        vx::launch_unit_blocks(ubase.as<uint32_t>(), ntri, (uint32_t)tot, btri.as<uint32_t>(), s);
    }
    return VX_OK;
}

// word_prefix + occupied count.  Split in two so that callers can queue dependent kernels before the host waits for the
// count (a host sync in front of a kernel leaves the GPU idle and the clocks down for its start).
vx_status prefix_launch(vx_grid* g, bool* pending, unsigned long long tag = 0, bool* tagged = nullptr)
{
    *pending = !g->occupied_known;
    if (tagged) *tagged = false;
    if (g->prefix_valid) return VX_OK;
    if (!tag) {  // a scan outside a build: its own sequence tag
        g->mail_seq = (g->mail_seq % 0xFFFFu) + 1u;
        tag = (unsigned long long)g->mail_seq << 48;
    }
    VX_HIP(g->wprefix.ensure((size_t)(g->g.nwords + 2) * 4));
    VX_HIP(g->wsel.ensure((size_t)(g->g.nwords / 32 + 4) * 4));  // at most 32 nwords / 1024 chunks of 1024 records
    VX_HIP(ensure_scan_tmp(g->scantmp, vx::scan_tmp_bytes(g->g.nwords), g->stream));
    VX_HIP(g->wp16.ensure((size_t)(g->g.nwords / 16 + 4) * 4));
    const bool tg = vx::launch_scan_u32(g->words.as<uint32_t>(), g->wprefix.as<uint32_t>(), g->g.nwords, true, g->scantmp.p, &g->mail->occupied, g->stream, true, tag,
                                        g->wsel.as<uint32_t>(), next_scan_gen(g->scantmp, g->stream), g->wp16.as<uint32_t>());
    g->sel_valid = tg;  // (the three-pass scan writes neither wsel nor wp16)
    if (tagged) *tagged = tg;
    g->occ_tag = tg ? tag : 0;  // what the host may poll the mailbox for instead of draining the stream (prefix_finish)
    g->prefix_valid = true;
    g->occupied_known = false;
    *pending = true;
    return VX_OK;
}

vx_status prefix_finish(vx_grid* g, bool pending)
{
    if (!pending || g->occupied_known) return VX_OK;
    static const bool poll = !(getenv("VOXHIP_POLL_MAIL") && atoi(getenv("VOXHIP_POLL_MAIL")) == 0);
    if (!(poll && g->occ_tag && mail_wait(&g->mail->occupied, nullptr, g->occ_tag, 5.0))) VX_HIP(hipStreamSynchronize(g->stream));
    const unsigned long long tot = g->mail->occupied & kMailValue;
    if (tot >= 0xFFFFFFFFull) return fail(VX_ERR_CAPACITY, "more than 2^32 occupied voxels");
    g->occupied = tot;
    g->occupied_known = true;
    return VX_OK;
}

vx_status ensure_prefix(vx_grid* g)
{
    DeviceGuard dg(g->device);
    bool pending = false;
    VX_TRY(prefix_launch(g, &pending));
    return prefix_finish(g, pending);
}

// from_tiled: the reference's bitmask has not been written yet -- the brick kernel reads the tiled build mask (g->twords) and writes it on the way
vx_status ensure_coarse(vx_grid* g, bool from_tiled = false)
{
    if (g->coarse_valid && !from_tiled) return VX_OK;
    DeviceGuard dg(g->device);
    for (int a = 0; a < 3; ++a) {
        g->cdim[a] = (g->g.dim[a] + vx::kCoarse - 1) / vx::kCoarse;
        g->c2dim[a] = (g->cdim[a] + vx::kCoarse - 1) / vx::kCoarse;
    }
    const uint64_t nc = (uint64_t)g->cdim[0] * g->cdim[1] * g->cdim[2];
    const uint64_t nc2 = (uint64_t)g->c2dim[0] * g->c2dim[1] * g->c2dim[2];
    VX_HIP(g->cwords.ensure((size_t)((nc + 63) / 64 * 2 + 2) * 4));
    VX_HIP(g->c2words.ensure((size_t)((nc2 + 31) / 32 + 2) * 4));
    VX_HIP(g->bricks.ensure((size_t)(nc * 8 * 3 + 8) * 8));  // three orientations (x, y, z slabs)
    // bitmask -> brick-major slabs in three orientations -> level-1 mip (from the z orientation) -> level-2 mip
    const bool fuse_mip1 = !(getenv("VOXHIP_FUSE_MIP1") && atoi(getenv("VOXHIP_FUSE_MIP1")) == 0);  // 0: the separate kernel, every brick stored (tests)
    const bool fused = vx::launch_build_bricks3(g->words.as<uint32_t>(), g->g.dim, g->cdim, g->bricks.as<unsigned long long>(),
                                                fuse_mip1 ? g->cwords.as<uint32_t>() : nullptr, g->stream, from_tiled ? g->twords.as<uint32_t>() : nullptr);
    if (!fused) vx::launch_brick_mip1(g->bricks.as<unsigned long long>() + 2ull * nc * 8ull, nc, g->cwords.as<uint32_t>(), g->stream);
    vx::launch_build_mip2(g->cwords.as<uint32_t>(), g->cdim, g->c2dim, g->c2words.as<uint32_t>(), g->stream);
    g->coarse_valid = true;
    return VX_OK;
}

vx_status sync_counts(vx_grid* g)
{
    if (g->counts_valid) return VX_OK;
    DeviceGuard dg(g->device);
    unsigned long long part[vx::kCallCounters * 8];
    VX_HIP(hipMemcpyAsync(part, g->small.as<Small>()->set_calls, sizeof(part), hipMemcpyDeviceToHost, g->stream));
    VX_HIP(hipStreamSynchronize(g->stream));
    unsigned long long sc = 0;
    for (uint32_t i = 0; i < vx::kCallCounters; ++i) sc += part[8 * i];
    g->set_calls = sc + g->host_set_calls;
    g->counts_valid = true;
    return VX_OK;
}

// `clear`: zero the bitmask and the call counter now (false: the caller queues that itself)
vx_status init_grid_storage(vx_grid* g, bool clear = true)
{
    VX_HIP(ensure_small(g->small));
    if (!g->mail) VX_HIP(mail_alloc(&g->mail));
    VX_HIP(g->words.ensure((size_t)(g->g.nwords + 2) * 4));
    if (clear) {
        VX_HIP(hipMemsetAsync(g->words.p, 0, (size_t)(g->g.nwords + 2) * 4, g->stream));
        VX_HIP(hipMemsetAsync(g->small.as<Small>()->set_calls, 0, sizeof(Small::set_calls), g->stream));
    }
    g->coarse_valid = g->prefix_valid = g->occupied_known = false;
    g->counts_valid = true;
    g->occupied = g->set_calls = g->host_set_calls = g->vec_count = 0;
    return VX_OK;
}

// Second half of a VX_VOXELIZE_MATERIALS build: addMatrialIfNeeded (voxelgrid.hpp:102-114) gives a material the next index when the
// first setVoxel call carrying it arrives, i.e. values are numbered by the first triangle that uses them; then triangle -> value ->
// index for every voxel (Bool / AABBstruct: the last triangle per voxel) or call (Vec).
vx_status finish_materials(vx_grid* g, const long long* first_use, size_t nvalues)
{
    if (!g->mat_pending) return VX_OK;
    if (nvalues != g->mat_values.size()) return fail(VX_ERR_INVALID_ARG, "first-use array does not match the mesh's material values");
    DeviceGuard dg(g->device);
    hipStream_t s = g->stream;
    std::vector<size_t> used;
    for (size_t v = 0; v < nvalues; ++v)
        if (first_use[v] >= 0) used.push_back(v);
    std::sort(used.begin(), used.end(), [&](size_t a, size_t b) { return first_use[a] < first_use[b]; });
    std::vector<int16_t> vindex(nvalues, (int16_t)-1);
    g->materials.clear();
    for (size_t k = 0; k < used.size(); ++k) {
        vindex[used[k]] = (int16_t)k;
        g->materials.push_back(g->mat_values[used[k]]);
    }
    const uint32_t ntri = g->mat_ntri;
    const bool vec = g->kind == VX_GRID_VEC;
    uint8_t* base = g->mattmp.as<uint8_t>();
    uint32_t* last_tri = vec ? nullptr : reinterpret_cast<uint32_t*>(base);
    uint8_t* tri_hit = base + (vec ? 0 : (size_t)g->mat_nids * 4);
    int16_t* value_index = reinterpret_cast<int16_t*>(tri_hit + (((size_t)ntri + 63) & ~(size_t)63));
    VX_HIP(hipMemcpyAsync(value_index, vindex.data(), vindex.size() * 2, hipMemcpyHostToDevice, s));
    VX_HIP(g->matids.ensure((size_t)g->mat_nids * 2 + 16));
    if (g->mat_nids) {
        if (vec)
            vx::launch_mat_ids_calls(g->recs.as<vx::TriRec>(), g->ubase.as<uint32_t>(), g->btri.as<uint32_t>(), ntri, g->umask.as<uint32_t>(), g->hbase.as<uint32_t>(),
                                     g->mat_dtv, value_index, g->matids.as<int16_t>(), s);
        else
            vx::launch_mat_ids(last_tri, g->mat_nids, g->mat_dtv, value_index, g->matids.as<int16_t>(), s);
    }
    VX_HIP(hipStreamSynchronize(s));  // vindex lives on this stack frame
    g->mat_count = g->mat_nids;
    g->has_materials = true;
    g->mat_pending = false;
    return VX_OK;
}

}  // namespace

// ============================================================================================================
extern "C" {

const char* vx_last_error(void) { return g_err.c_str(); }

const char* vx_status_string(vx_status s)
{
    switch (s) {
        case VX_OK: return "ok";
        case VX_ERR_INVALID_ARG: return "invalid argument";
        case VX_ERR_PATH: return "Path does not exist!";
        case VX_ERR_PARSE: return "Colud not get valid reader!";
        case VX_ERR_OUT_OF_BOUNDS: return "Index out of bounds";
        case VX_ERR_MORTON_BITS: return "We support up to 21 bits per axis (max 2^21 voxels per dimension)!";
        case VX_ERR_NO_DEVICE: return "no HIP device";
        case VX_ERR_HIP: return "HIP runtime error";
        case VX_ERR_CAPACITY: return "capacity exceeded";
        case VX_ERR_UNSUPPORTED: return "unsupported";
    }
    return "unknown";
}

int vx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

vx_status vx_set_device(int device)
{
    VX_TRY(need_device(device));
    g_device = device;
    return VX_OK;
}

vx_status vx_release_cached_memory(void)
{
    int n = vx_device_count();
    for (int d = 0; d < n && d < kMaxDev; ++d) {
        std::vector<FreeBlock> blocks;
        {
            std::lock_guard<std::mutex> lk(g_pool[d].mu);
            for (auto& kv : g_pool[d].free_blocks) blocks.push_back(kv.second);
            g_pool[d].free_blocks.clear();
        }
        if (blocks.empty()) continue;
        DeviceGuard dg(d);
        for (FreeBlock& fb : blocks) {
            if (fb.ev) (void)hipEventSynchronize(fb.ev);  // hipFree of a block that queued kernels still use would be the same race
            (void)hipFree(fb.p);
            if (fb.ev) { std::lock_guard<std::mutex> lk(g_pool[d].mu); g_pool[d].spare_events.push_back(fb.ev); }
        }
    }
    return VX_OK;
}

// ---- mesh ---------------------------------------------------------------------------------------------------
vx_status vx_mesh_load_obj(const char* path, vx_mesh** out)
{
    if (!path || !out) return fail(VX_ERR_INVALID_ARG, "null argument");
    vx_mesh* m = new vx_mesh();
    std::string msg;
    const int rc = vx::load_obj(path, m->hv, m->hi, m->tri_mat, m->materials, msg);
    if (rc == 1) { delete m; return fail(VX_ERR_PATH, "Path does not exist!"); }                                    // VoxelBuilder.hpp:54-56
    if (rc == 2) { delete m; return fail(VX_ERR_PARSE, "Colud not get valid reader! Error message " + msg); }        // :63-65
    m->nv = m->hv.size() / 3;
    m->nt = m->hi.size() / 3;
    if (m->materials.empty()) m->tri_mat.clear();  // no mtllib: every id is -1
    m->device = g_device;
    *out = m;
    return VX_OK;
}

vx_status vx_mesh_from_arrays(const float* xyz, size_t nv, const int32_t* idx, size_t nt, vx_mesh** out)
{
    if (!out || (nv && !xyz) || (nt && !idx)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (nv >= 0xFFFFFFFFull || nt >= 0xFFFFFFFEull) return fail(VX_ERR_CAPACITY, "mesh too large for 32-bit indices");
    for (size_t i = 0; i < nt * 3; ++i)
        if (idx[i] < 0 || (size_t)idx[i] >= nv) return fail(VX_ERR_INVALID_ARG, "triangle index out of range");
    vx_mesh* m = new vx_mesh();
    m->hv.assign(xyz, xyz + nv * 3);
    m->hi.assign(idx, idx + nt * 3);
    m->nv = nv;
    m->nt = nt;
    m->device = g_device;
    *out = m;
    return VX_OK;
}

vx_status vx_mesh_from_device(const float* dxyz, size_t nv, const int32_t* didx, size_t nt, vx_mesh** out)
{
    if (!out || (nv && !dxyz) || (nt && !didx)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (nv >= 0xFFFFFFFFull || nt >= 0xFFFFFFFEull) return fail(VX_ERR_CAPACITY, "mesh too large for 32-bit indices");
    VX_TRY(need_device(g_device));
    vx_mesh* m = new vx_mesh();
    m->nv = nv;
    m->nt = nt;
    m->dv = dxyz;
    m->di = didx;
    m->borrowed = true;
    m->device = g_device;
    *out = m;
    return VX_OK;
}

size_t vx_mesh_num_vertices(const vx_mesh* m) { return m ? m->nv : 0; }
size_t vx_mesh_num_triangles(const vx_mesh* m) { return m ? m->nt : 0; }
const float* vx_mesh_host_vertices(const vx_mesh* m) { return (m && !m->borrowed) ? m->hv.data() : nullptr; }
const int32_t* vx_mesh_host_indices(const vx_mesh* m) { return (m && !m->borrowed) ? m->hi.data() : nullptr; }
size_t vx_mesh_num_materials(const vx_mesh* m) { return m ? m->materials.size() : 0; }
vx_status vx_mesh_materials(const vx_mesh* m, vx_material* out, size_t cap)
{
    if (!m || (!out && cap)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (cap < m->materials.size()) return fail(VX_ERR_CAPACITY, "material buffer too small");
    if (!m->materials.empty()) std::memcpy(out, m->materials.data(), m->materials.size() * sizeof(vx_material));
    return VX_OK;
}
const int32_t* vx_mesh_host_material_ids(const vx_mesh* m) { return (m && !m->tri_mat.empty()) ? m->tri_mat.data() : nullptr; }
vx_status vx_mesh_set_materials(vx_mesh* m, const vx_material* mats, size_t n, const int32_t* ids)
{
    if (!m || (n && !mats)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (n > 32767) return fail(VX_ERR_CAPACITY, "more than 32767 materials: per-voxel ids are int16 (voxelgrid.hpp:29)");
    if (ids)
        for (size_t t = 0; t < m->nt; ++t)
            if (ids[t] < -1 || (ids[t] >= 0 && (size_t)ids[t] >= n)) return fail(VX_ERR_INVALID_ARG, "material id out of range");
    m->materials.assign(mats, mats + n);
    if (ids && n) m->tri_mat.assign(ids, ids + m->nt); else m->tri_mat.clear();
    m->values_ready = false;
    return VX_OK;
}

void vx_mesh_free(vx_mesh* m)
{
    if (!m) return;
    if (m->uploaded || m->btv.p) {
        // grids on any stream may still be reading the vertex / index arrays: wait for the device before the blocks go back
        DeviceGuard dg(m->device);
        (void)hipDeviceSynchronize();
    }
    m->bv.release(/*in_flight=*/false);
    m->bi.release(/*in_flight=*/false);
    m->btv.release(/*in_flight=*/false);
    delete m;
}

// ---- voxelize -------------------------------------------------------------------------------------------------
vx_status vx_voxelize_into(const vx_mesh* mesh_c, float vs, const vx_voxelize_opts* opts, vx_grid* g)
{
    if (!mesh_c || !g) return fail(VX_ERR_INVALID_ARG, "null argument");
    VX_TRY(check_voxel_size(vs));
    vx_mesh* mesh = const_cast<vx_mesh*>(mesh_c);
    VX_TRY(need_device(g->device));
    if (mesh->device != g->device) return fail(VX_ERR_INVALID_ARG, "mesh and grid live on different devices");
    DeviceGuard dg(g->device);
    VX_TRY(mesh_to_device(mesh));
    vx_voxelize_opts o{};
    if (opts) o = *opts;
    g->set_stream((hipStream_t)o.stream);
    hipStream_t s = g->stream;
    // a list emission of the previous build that is still to come reads the unit masks, hit bases and records this build overwrites (one
    // that was never queued is dropped: its list is about to be replaced)
    VX_HIP(g->list_resolve(/*drop_unqueued=*/true));
    if (o.sat_variant != 0 && o.sat_variant != 1) return fail(VX_ERR_INVALID_ARG, "sat_variant must be 0 or 1");
    const bool want_mat = (o.flags & VX_VOXELIZE_MATERIALS) != 0;
    const bool list_async = (o.flags & VX_VOXELIZE_LIST_ASYNC) != 0 && g->kind == VX_GRID_VEC && !want_mat;  // (the material ids need the hit bases on the main stream)
    if (want_mat) VX_TRY(mesh_material_values(mesh));
    g->has_materials = false;
    g->mat_pending = false;
    g->mat_gathered = false;
    g->materials.clear();
    g->mat_count = 0;
    if (o.shard_world < 0 || (o.shard_world > 0 && (o.shard_rank < 0 || o.shard_rank >= o.shard_world))) return fail(VX_ERR_INVALID_ARG, "shard_rank / shard_world out of range");
    const bool by_rank = o.shard_world > 1 && !(o.word_begin || o.word_end);

    VX_HIP(ensure_small(g->small));
    if (!g->mail) VX_HIP(mail_alloc(&g->mail));
    Small* ds = g->small.as<Small>();
    uint64_t tb = 0, te = mesh->nt;
    if (o.tri_begin || o.tri_end) {
        if (o.tri_begin > o.tri_end || o.tri_end > mesh->nt) return fail(VX_ERR_INVALID_ARG, "triangle shard out of range");
        tb = o.tri_begin;
        te = o.tri_end;
    }
    const uint32_t ntri = (uint32_t)(te - tb);
    const bool sharded_words = o.word_begin || o.word_end || by_rank;

    // Unsharded build: K1 leaves origin + dims in device memory, so the triangle records and the unit scan are queued right
    // behind it and the host waits ONCE for the bbox and the unit count (every host round trip costs ~20 us of idle GPU: the
    // wake-up plus the launch latency of an empty queue).  The bitmask of the previous build is cleared in the same window.
    // sequence tag of this build's totals in the mailbox (never 0: an untagged word never matches); VOXHIP_POLL_MAIL=0: the host
    // drains the stream instead of polling the mailbox
    g->mail_seq = (g->mail_seq % 0xFFFFu) + 1u;
    const unsigned long long mtag = (unsigned long long)g->mail_seq << 48;
    static const bool poll_mail = !(getenv("VOXHIP_POLL_MAIL") && atoi(getenv("VOXHIP_POLL_MAIL")) == 0);
    Extent ex;
    uint64_t btri_entries = 0;  // entries of the block table filled by the launch queued ahead of the unit total
    bool setup_queued = false;
    size_t cleared = 0;         // bytes of the build's mask cleared ahead of the host wait ...
    bool cleared_tiled = false;  // ... of the tiled build mask, that is
    if (ntri > 0) {  // (a word shard's z slab is derived from the device-side dims by the record kernel itself)
        vx::GridParams gp{};
        const float zero3[3] = {0.f, 0.f, 0.f};
        const uint64_t zdim[3] = {0, 0, 0};
        fill_params(gp, zero3, vs, zdim);
        vx::launch_bbox(mesh->dv, mesh->nv, ds->bbox_state, g->mail->bbox, ds->set_calls, s, vs, &ds->dgrid);
        bool units_tagged = false;
        // the previous build's bitmask is cleared in the same window -- by the record kernel's own threads when there are enough of
        // them (at most 256 bytes each), by a memset behind the block table otherwise
        // (a handle whose previous build went through the tiled build mask is expected to do so again: that is the buffer to clear)
        DevBuf& cb = (g->last_tiled && g->twords.p) ? g->twords : g->words;
        cleared_tiled = &cb == &g->twords;
        const bool clear_in_setup = cb.p && (cb.cap % 16) == 0 && cb.cap / 256 <= (size_t)ntri;
        VX_TRY(setup_launch(mesh, gp, o.sat_variant, tb, ntri, 0, 0, g->recs, g->units, g->ubase, g->scantmp, g->mail, s, g->ext, &ds->dgrid, mtag, &units_tagged,
                            clear_in_setup ? cb.p : nullptr, clear_in_setup ? cb.cap : 0, sharded_words ? o.word_begin : 0,
                            sharded_words ? o.word_end : 0, by_rank ? (uint32_t)o.shard_rank : 0u, by_rank ? (uint32_t)o.shard_world : 0u));
        if (clear_in_setup) cleared = cb.cap;
        // the block table of the units: queued now, for as many blocks as the handle's table from the previous build holds, so that
        // it runs while the host waits for the unit total (redone by setup_finish should the table turn out too small)
        if (g->btri.p && g->btri.cap >= 64) {
            btri_entries = g->btri.cap / 4;
            if (btri_entries > 0x3FFFFFFull) btri_entries = 0x3FFFFFFull;
            vx::launch_unit_blocks(g->ubase.as<uint32_t>(), ntri, (uint32_t)((btri_entries - 2) * 64), g->btri.as<uint32_t>(), s, (uint32_t)btri_entries);
        }
        if (cb.p && !clear_in_setup) {
            VX_HIP(hipMemsetAsync(cb.p, 0, cb.cap, s));
            cleared = cb.cap;
        }
        // the bbox (written by k_bbox, two kernels earlier) and the unit total: polled from the mailbox, see the hit count below
        if (!(poll_mail && units_tagged && mail_wait(&g->mail->units, nullptr, mtag, 5.0))) VX_HIP(hipStreamSynchronize(s));
        VX_TRY(extent_from_bbox(g->mail->bbox, mesh->nv, vs, &ex));
        setup_queued = true;
    } else {
        VX_TRY(compute_extent(mesh, vs, ds, g->mail, s, &ex));  // also clears the per-build call counter
    }
    const uint64_t nvox = ex.dim[0] * ex.dim[1] * ex.dim[2];
    if (nvox > kMaxVoxels) return fail(VX_ERR_CAPACITY, "grid exceeds 2^37 voxels");
    fill_params(g->g, ex.mn, vs, ex.dim);
    for (int a = 0; a < 3; ++a) { g->bbmin[a] = ex.mn[a]; g->bbmax[a] = ex.mx[a]; g->bbc[a] = ex.ctr[a]; }
    VX_TRY(init_grid_storage(g, /*clear=*/false));
    const size_t mask_bytes = (size_t)(g->g.nwords + 2) * 4;
    const bool words_fresh = g->words.fresh;
    if (g->words.fresh) { if (!cleared_tiled) cleared = 0; g->words.fresh = false; }  // a new block: the early clear hit the old one
    bool mask_is_clear = !cleared_tiled && cleared >= mask_bytes;

    uint64_t wb = 0, we = g->g.nwords;
    if (by_rank) {
        vx_shard_words(g->g.nwords, o.shard_rank, o.shard_world, &wb, &we, nullptr);
    } else if (sharded_words) {
        if (o.word_begin > o.word_end || o.word_end > g->g.nwords) return fail(VX_ERR_INVALID_ARG, "word shard out of range");
        wb = o.word_begin;
        we = o.word_end;
    }
    const bool whole_build = wb == 0 && we == g->g.nwords && tb == 0 && te == mesh->nt;
    auto no_calls_materials = [&]() -> vx_status {  // a build without a setVoxel call: no material is ever used
        if (!want_mat) return VX_OK;
        g->mat_pending = true;
        g->mat_values = mesh->values;
        g->mat_first_use.assign(mesh->values.size(), -1);
        g->mat_dtv = nullptr; g->mat_ntri = 0; g->mat_nids = 0;
        VX_HIP(g->mattmp.ensure(mesh->values.size() * 2 + 256));
        if (whole_build) return finish_materials(g, g->mat_first_use.data(), g->mat_first_use.size());
        return VX_OK;
    };
    g->triangles = te - tb;
    if (ntri == 0 || nvox == 0 || wb == we) {
        if (!mask_is_clear) VX_HIP(hipMemsetAsync(g->words.p, 0, mask_bytes, s));
        return no_calls_materials();
    }

    // Unsharded words, rows of whole words: the voxelizer ORs into the tiled build mask (one atomic request per 4 x 4 rows instead of one per
    // row, launch_voxelize) and launch_untile writes every word of the reference's bitmask from it.  VOXHIP_VOX_TILED=0: always the direct form.
    static const bool tiled_ok = !(getenv("VOXHIP_VOX_TILED") && atoi(getenv("VOXHIP_VOX_TILED")) == 0);
    static const bool tiled_shards = !(getenv("VOXHIP_VOX_TILED_SHARDS") && atoi(getenv("VOXHIP_VOX_TILED_SHARDS")) == 0);
    const bool whole_words = wb == 0 && we == g->g.nwords;
    // (a word shard goes the same way: the voxelizer keeps to the rows whose words it owns, launch_untile writes those and zeroes the rest --
    // a second pass over the WHOLE mask per rank, so only where the mask is small against the mesh: the size rule of the clear that rides in
    // the record kernel's threads.  atrium262k, two logical ranks on one GPU: 512^3 voxelize stage 121 -> 95 us; 1024^3 392 -> 426, hence the rule)
    // ... and only for shards of a fifth of the mask or more: shard 0 of 2 / 4 / 8 at 512^3 (tools/shard_time.py, kernels of a rebuild)
    // 119.7 -> 97.9 / 90.6 -> 73.3 / 55.1 -> 64.0 us -- at an eighth the un-tiling pass costs what the voxelizer no longer has to gain
    const bool shard_small = (size_t)vx::tiled_mask_words(g->g.dim) * 4 / 256 <= (size_t)ntri && (we - wb) * 5 >= g->g.nwords;
    const bool tiled = tiled_ok && (whole_words || (tiled_shards && shard_small)) && (ex.dim[0] % 32) == 0;
    g->last_tiled = tiled;
    if (tiled) {
        const size_t tbytes = (size_t)vx::tiled_mask_words(g->g.dim) * 4;
        VX_HIP(g->twords.ensure(tbytes));
        const bool tw_clear = cleared_tiled && cleared >= tbytes && !g->twords.fresh;
        g->twords.fresh = false;
        if (!tw_clear) VX_HIP(hipMemsetAsync(g->twords.p, 0, tbytes, s));
        if (words_fresh) VX_HIP(hipMemsetAsync(g->words.p, 0, mask_bytes, s));  // (the two words behind the mask)
        mask_is_clear = true;  // every word of the mask is written by launch_untile
    }
    uint64_t U = 0;
    if (setup_queued) {
        if (!mask_is_clear) VX_HIP(hipMemsetAsync(g->words.p, 0, mask_bytes, s));
        VX_TRY(setup_finish(ntri, g->ubase, g->btri, g->mail, s, &U, /*stream_is_drained=*/true, btri_entries));
    } else {
        // z slab that contains the voxels of words [wb, we)
        const uint64_t XY = ex.dim[0] * ex.dim[1];
        uint32_t zlo = (uint32_t)((wb * 32) / XY);
        uint64_t zh = (we * 32 + XY - 1) / XY;
        if (zh > ex.dim[2]) zh = ex.dim[2];
        const uint32_t zhi = (uint32_t)zh;
        // records + unit scan are queued, THEN the bitmask is cleared (it does not depend on the unit count), then the host waits
        VX_TRY(setup_launch(mesh, g->g, o.sat_variant, tb, ntri, zlo, zhi, g->recs, g->units, g->ubase, g->scantmp, g->mail, s, g->ext));
        VX_HIP(hipMemsetAsync(g->words.p, 0, mask_bytes, s));
        VX_TRY(setup_finish(ntri, g->ubase, g->btri, g->mail, s, &U));
    }
    if (U == 0) return no_calls_materials();
    uint32_t* umask = nullptr;
    if (g->kind == VX_GRID_VEC || want_mat) {
        VX_HIP(g->umask.ensure((size_t)(U + 1) * 4));
        umask = g->umask.as<uint32_t>();
    }
    // an axis above 65535 cells: the unit kernels also read the extension words of the candidate ranges
    const uint32_t* xw = (ex.dim[0] > 65535 || ex.dim[1] > 65535 || ex.dim[2] > 65535) ? g->ext.as<uint32_t>() : nullptr;
    // VoxelGridVec::setVoxel appends one Aabb per call (voxelgridVecEncoding.cpp:19-39): ordered emission, at the exclusive scan of the
    // hit counts -- two-level: the voxelizer leaves the hits of every block of 64 units, the device scan runs over those
    const uint64_t nUB = (U + 63) / 64;
    uint32_t* bhits = nullptr;
    if (g->kind == VX_GRID_VEC) {
        VX_HIP(g->bhits.ensure((size_t)(nUB + 4) * 4));
        VX_HIP(g->hbase.ensure((size_t)(nUB + 4) * 4));
        bhits = g->bhits.as<uint32_t>();
    }
    vx::launch_voxelize(g->recs.as<vx::TriRec>(), g->ubase.as<uint32_t>(), g->btri.as<uint32_t>(), ntri, g->g, o.sat_variant,
                        tiled ? g->twords.as<uint32_t>() : g->words.as<uint32_t>(), wb, we, umask, ds->set_calls, s, xw, bhits, tiled);
    // (the tiled mask -> the reference's bitmask: by the brick kernel on its way when the traversal structure is built right away, below)
    static const bool eager = !(getenv("VOXHIP_EAGER") && atoi(getenv("VOXHIP_EAGER")) == 0);
    static const bool fuse_untile = !(getenv("VOXHIP_FUSE_UNTILE") && atoi(getenv("VOXHIP_FUSE_UNTILE")) == 0);
    const bool untile_in_bricks = tiled && eager && fuse_untile && whole_words;
    if (tiled && !untile_in_bricks) vx::launch_untile(g->twords.as<uint32_t>(), g->words.as<uint32_t>(), g->g.dim, s, wb, we);
    g->counts_valid = false;
    bool hits_tagged = false, occ_tagged = false, occ_queued = false;
    if (g->kind == VX_GRID_VEC) {
        VX_HIP(ensure_scan_tmp(g->scantmp, vx::scan_tmp_bytes(nUB), s));
        hits_tagged = vx::launch_scan_u32(bhits, g->hbase.as<uint32_t>(), nUB, false, g->scantmp.p, &g->mail->hits, s, true, mtag, nullptr, next_scan_gen(g->scantmp, s));
    }
    // A complete (unsharded) bitmask: queue what every consumer of the grid needs next -- the traversal structure (bricks,
    // bounds, mips = the reference's acceleration-structure build, hello_vulkan.cpp:700-703) and the word prefix (getAabbs /
    // primitive ids) -- behind the voxelizer instead of lazily in front of the first query.  For the Vec flavour this work
    // runs while the host waits for the hit count.  VOXHIP_EAGER=0 keeps it lazy.
    if (eager && wb == 0 && we == g->g.nwords) {
        VX_TRY(ensure_coarse(g, untile_in_bricks));
        bool pending = false;
        occ_queued = !g->prefix_valid;
        VX_TRY(prefix_launch(g, &pending, mtag, &occ_tagged));
    }
    if (g->kind == VX_GRID_VEC) {
        // The list is emitted into the handle's existing buffer before the host knows the hit count (writes beyond the
        // buffer's capacity are dropped by the kernel); only a list that outgrew it is emitted again after the wait.
        const bool to_bound = g->bound != nullptr && g->bound_cap > 0;
        vx_aabb* tgt = to_bound ? g->bound : g->vec.as<vx_aabb>();
        const uint64_t cap_rec = to_bound ? g->bound_cap + 1 : (g->vec.p ? g->vec.cap / sizeof(vx_aabb) : 0);
        g->vec_in_bound = false;
        if (cap_rec && !list_async)
            vx::launch_emit_units(g->recs.as<vx::TriRec>(), g->ubase.as<uint32_t>(), g->btri.as<uint32_t>(), ntri, g->g, umask,
                                  g->hbase.as<uint32_t>(), tgt, nullptr, s, to_bound ? g->bound_cap : cap_rec, xw);
        // The host needs the hit count (and takes the occupied count along).  Both are written by scans that run BEFORE the
        // emission: the host polls the tagged mailbox words and goes on queueing work (the caller's next call: a trace) while
        // the emission still runs; a stream synchronize would wake it ~15 us after the last kernel.  VOXHIP_POLL_MAIL=0, an
        // untagged total (three-pass scan) or 5 ms without an answer: the synchronize.
        // VX_VOXELIZE_LIST_ASYNC: only the hit count -- its scan runs in front of the traversal structure and the word prefix, so the host
        // is back in the caller ~40 us of GPU work before the build ends and the caller's ray batch is queued in time; the occupied count
        // (the LAST kernel's total) is fetched when somebody asks for it (prefix_finish).
        const bool occ_in_flight = g->prefix_valid && !g->occupied_known;
        const bool occ_along = occ_in_flight && !(list_async && occ_queued && occ_tagged);
        bool got = false;
        if (poll_mail && hits_tagged && (!occ_along || (occ_queued && occ_tagged)))
            got = mail_wait(&g->mail->hits, occ_along ? &g->mail->occupied : nullptr, mtag, 5.0);
        if (!got) VX_HIP(hipStreamSynchronize(s));
        const unsigned long long hits = g->mail->hits & kMailValue;
        if (hits >= 0xFFFFFFFFull) return fail(VX_ERR_CAPACITY, "more than 2^32 voxel hits");
        if (g->prefix_valid && (occ_along || !got)) {  // the same wait covered the occupied count
            if ((g->mail->occupied & kMailValue) >= 0xFFFFFFFFull) return fail(VX_ERR_CAPACITY, "more than 2^32 occupied voxels");
            g->occupied = g->mail->occupied & kMailValue;
            g->occupied_known = true;
        }
        if (hits + 1 <= cap_rec) g->vec_in_bound = to_bound;
        if (hits + 1 > cap_rec) VX_HIP(g->vec.ensure((size_t)(hits + 1) * sizeof(vx_aabb)));
        if (list_async) {
            // VX_VOXELIZE_LIST_ASYNC: the count is known, the records are written later -- beside the next ray batch (trace_common), or
            // on this stream by whatever asks for them first (list_resolve)
            g->ld.from_mask = false;
            g->ld.ntri = ntri;
            g->ld.ext = xw != nullptr;
            g->ld.tgt = g->vec_ptr();
            g->ld.cap = g->vec_in_bound ? g->bound_cap : ~0ull;
            g->list_deferred = hits != 0;
        } else if (hits + 1 > cap_rec) {
            vx::launch_emit_units(g->recs.as<vx::TriRec>(), g->ubase.as<uint32_t>(), g->btri.as<uint32_t>(), ntri, g->g, umask,
                                  g->hbase.as<uint32_t>(), g->vec.as<vx_aabb>(), nullptr, s, ~0ull, xw);
        }
        g->vec_count = hits;
    }
    if (want_mat) {
        // ---- per-voxel material ids (see k_mat_last): needs the word prefix (queued above for an unsharded build) and, for the Vec
        // flavour, the hit bases.  First half here: the last triangle per voxel of this shard and, per material value, the first
        // triangle of this shard that uses it; second half (finish_materials) once the first uses of ALL shards are known -- at once
        // for a whole build.
        bool pending = false;
        VX_TRY(prefix_launch(g, &pending));
        VX_TRY(prefix_finish(g, pending));
        const uint64_t nids = g->kind == VX_GRID_VEC ? g->vec_count : g->occupied;
        const size_t tmp_bytes = (g->kind == VX_GRID_VEC ? 0 : (size_t)nids * 4) + (size_t)ntri + 64 + mesh->values.size() * 2 + 64;
        VX_HIP(g->mattmp.ensure(tmp_bytes));
        uint8_t* base = g->mattmp.as<uint8_t>();
        uint32_t* last_tri = g->kind == VX_GRID_VEC ? nullptr : reinterpret_cast<uint32_t*>(base);
        uint8_t* tri_hit = base + (g->kind == VX_GRID_VEC ? 0 : (size_t)nids * 4);
        VX_HIP(hipMemsetAsync(base, 0, (size_t)(tri_hit - base) + ntri, s));
        vx::launch_mat_last(g->recs.as<vx::TriRec>(), g->ubase.as<uint32_t>(), g->btri.as<uint32_t>(), ntri, g->g, umask, g->words.as<uint32_t>(),
                            g->wprefix.as<uint32_t>(), last_tri, tri_hit, s, xw);
        std::vector<uint8_t> hit(ntri);
        VX_HIP(hipMemcpyAsync(hit.data(), tri_hit, ntri, hipMemcpyDeviceToHost, s));
        VX_HIP(hipStreamSynchronize(s));
        g->mat_values = mesh->values;
        g->mat_first_use.assign(mesh->values.size(), -1);
        const int32_t* tv = mesh->tri_value.data() + tb;
        for (uint32_t t = 0; t < ntri; ++t)
            if (hit[t] && g->mat_first_use[(size_t)tv[t]] < 0) g->mat_first_use[(size_t)tv[t]] = (long long)(tb + t);
        g->mat_dtv = mesh->btv.as<int32_t>() + tb;
        g->mat_ntri = ntri;
        g->mat_nids = nids;
        g->mat_pending = true;
        if (whole_build) VX_TRY(finish_materials(g, g->mat_first_use.data(), g->mat_first_use.size()));
    }
    VX_HIP(hipGetLastError());
    return VX_OK;
}

vx_status vx_voxelize(const vx_mesh* mesh, float vs, vx_grid_kind kind, const vx_voxelize_opts* opts, vx_grid** out)
{
    if (!mesh || !out) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (kind != VX_GRID_BOOL && kind != VX_GRID_AABBSTRUCT && kind != VX_GRID_VEC) return fail(VX_ERR_INVALID_ARG, "unknown grid kind");
    vx_grid* g = new vx_grid();
    g->kind = kind;
    g->set_dev(mesh->device);
    const vx_status s = vx_voxelize_into(mesh, vs, opts, g);
    if (s != VX_OK) {
        g->release_all();
        delete g;
        return s;
    }
    *out = g;
    return VX_OK;
}

// ---- multi-GPU build inside one process -----------------------------------------------------------------------------
// vx_multi: the persistent form.  Created once per (mesh, device list): the mesh is uploaded to every device, every rank gets a
// grid handle and a worker thread that lives as long as the context.  vx_multi_voxelize then runs with no upload, no allocation
// (the grids' buffers are reused when sizes repeat) and no thread start: rank k voxelizes the word shard vx_shard_words(.., k, n)
// that it derives from its own bounding-box pass (vx_voxelize_opts.shard_rank / shard_world: nobody needs the grid's word count
// beforehand), the shards travel as peer copies, the destination grids rebuild word prefix and traversal structure.
struct vx_multi {
    int nd = 0;
    vx_grid_kind kind = VX_GRID_BOOL;
    std::vector<int> devices;
    std::vector<vx_mesh*> meshes;
    std::vector<vx_grid*> grids;
    // workers
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t epoch = 0;       // bumped by vx_multi_voxelize: the workers' signal
    int remaining = 0;        // workers still busy with the current epoch
    bool quit = false;
    float vs = 0.f;
    vx_voxelize_opts opts{};
    std::vector<vx_status> st;
    std::vector<std::string> msg;
};

namespace {
void multi_worker(vx_multi* m, int k)
{
    g_device = m->devices[(size_t)k];  // thread-local
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(m->mu);
            m->cv_go.wait(lk, [&] { return m->quit || m->epoch != seen; });
            if (m->quit) return;
            seen = m->epoch;
        }
        vx_voxelize_opts o = m->opts;
        o.shard_rank = k;
        o.shard_world = m->nd;
        o.word_begin = o.word_end = o.tri_begin = o.tri_end = 0;
        const vx_status s = vx_voxelize_into(m->meshes[(size_t)k], m->vs, &o, m->grids[(size_t)k]);
        m->st[(size_t)k] = s;
        if (s != VX_OK) m->msg[(size_t)k] = g_err;
        else {  // the shard must be complete before another device copies it
            DeviceGuard dg(m->grids[(size_t)k]->device);
            (void)hipStreamSynchronize(m->grids[(size_t)k]->stream);
        }
        {
            std::lock_guard<std::mutex> lk(m->mu);
            if (--m->remaining == 0) m->cv_done.notify_all();
        }
    }
}
}  // namespace

vx_status vx_multi_create(const vx_mesh* mesh, const int* devices, int nd, vx_grid_kind kind, vx_multi** out)
{
    if (!mesh || !devices || !out || nd < 1) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (kind != VX_GRID_BOOL && kind != VX_GRID_AABBSTRUCT)
        return fail(VX_ERR_UNSUPPORTED, "multi-GPU builds are VX_GRID_BOOL / VX_GRID_AABBSTRUCT (VX_GRID_VEC's list order needs triangle shards)");
    if (mesh->borrowed) return fail(VX_ERR_UNSUPPORTED, "multi-GPU builds need a mesh with host arrays (vx_mesh_load_obj / vx_mesh_from_arrays)");
    for (int k = 0; k < nd; ++k) VX_TRY(need_device(devices[k]));
    vx_multi* m = new vx_multi();
    m->nd = nd;
    m->kind = kind;
    m->devices.assign(devices, devices + nd);
    m->meshes.assign((size_t)nd, nullptr);
    m->grids.assign((size_t)nd, nullptr);
    m->st.assign((size_t)nd, VX_OK);
    m->msg.assign((size_t)nd, std::string());
    const int prev_device = g_device;
    vx_status s = VX_OK;
    for (int k = 0; k < nd && s == VX_OK; ++k) {
        g_device = devices[k];
        s = vx_mesh_from_arrays(mesh->hv.data(), mesh->nv, mesh->hi.data(), mesh->nt, &m->meshes[(size_t)k]);
        if (s == VX_OK && !mesh->materials.empty())
            s = vx_mesh_set_materials(m->meshes[(size_t)k], mesh->materials.data(), mesh->materials.size(), mesh->tri_mat.empty() ? nullptr : mesh->tri_mat.data());
        if (s == VX_OK) s = mesh_to_device(m->meshes[(size_t)k]);
        if (s == VX_OK) {
            vx_grid* g = new vx_grid();
            g->kind = kind;
            g->set_dev(devices[k]);
            m->grids[(size_t)k] = g;
        }
    }
    g_device = prev_device;
    if (s != VX_OK) { const std::string e = g_err; vx_multi_free(m); return fail(s, e); }
    for (int k = 0; k < nd; ++k) m->threads.emplace_back(multi_worker, m, k);
    *out = m;
    return VX_OK;
}

vx_status vx_multi_voxelize(vx_multi* m, float vs, const vx_voxelize_opts* opts, int all_gather)
{
    if (!m) return fail(VX_ERR_INVALID_ARG, "null argument");
    VX_TRY(check_voxel_size(vs));
    vx_voxelize_opts o{};
    if (opts) o = *opts;
    if (o.word_begin || o.word_end || o.tri_begin || o.tri_end || o.shard_world) return fail(VX_ERR_INVALID_ARG, "vx_multi_voxelize shards the build itself");
    const bool want_mat = (o.flags & VX_VOXELIZE_MATERIALS) != 0;
    const int nd = m->nd;
    for (int k = 0; k < nd; ++k)
        if (!m->grids[(size_t)k]) return fail(VX_ERR_INVALID_ARG, "a grid of this context was handed to the caller (vx_multi_release_grid): create a new context");
    {   // ---- every rank builds its shard, side by side
        std::unique_lock<std::mutex> lk(m->mu);
        m->vs = vs;
        m->opts = o;
        m->remaining = nd;
        ++m->epoch;
        m->cv_go.notify_all();
        m->cv_done.wait(lk, [&] { return m->remaining == 0; });
    }
    for (int k = 0; k < nd; ++k)
        if (m->st[(size_t)k] != VX_OK) return fail(m->st[(size_t)k], m->msg[(size_t)k]);
    const uint64_t nwords = m->grids[0]->g.nwords;
    for (int k = 1; k < nd; ++k)
        if (m->grids[(size_t)k]->g.nwords != nwords) return fail(VX_ERR_HIP, "ranks disagree about the grid (different devices gave different bounding boxes?)");
    // ---- exchange: every destination pulls the other ranks' word ranges as peer copies (one slab per source device)
    const int ndst = all_gather ? nd : 1;
    hipError_t e = hipSuccess;
    for (int d = 0; d < ndst && e == hipSuccess; ++d) {
        vx_grid* gd = m->grids[(size_t)d];
        DeviceGuard dg(gd->device);
        for (int k = 0; k < nd && e == hipSuccess; ++k) {
            if (k == d) continue;
            vx_grid* gk = m->grids[(size_t)k];
            if (gk->device != gd->device) {  // direct xGMI copies instead of staging through the host
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, gd->device, gk->device) == hipSuccess && can) {
                    (void)hipDeviceEnablePeerAccess(gk->device, 0);
                    (void)hipGetLastError();  // (already enabled is not an error)
                }
            }
            uint64_t wb = 0, we = 0;
            vx_shard_words(nwords, k, nd, &wb, &we, nullptr);
            if (we <= wb) continue;
            e = hipMemcpyPeerAsync(gd->words.as<uint32_t>() + wb, gd->device, gk->words.as<uint32_t>() + wb, gk->device, (size_t)(we - wb) * 4, gd->stream);
        }
    }
    for (int d = 0; d < ndst && e == hipSuccess; ++d) {
        DeviceGuard dg(m->grids[(size_t)d]->device);
        e = hipStreamSynchronize(m->grids[(size_t)d]->stream);
    }
    if (e != hipSuccess) return fail(VX_ERR_HIP, std::string("peer exchange: ") + hipGetErrorString(e));
    // ---- materials: the index of a material is the order of its first use over ALL shards (vx_grid_finish_materials); the ids of the
    // voxels in ascending order are the shards' id arrays one after the other
    std::vector<uint64_t> shard_ids((size_t)nd, 0);
    if (want_mat) {
        const size_t nv = m->grids[0]->mat_first_use.size();
        std::vector<long long> fu(nv, -1);
        for (int k = 0; k < nd; ++k) {
            const std::vector<long long>& a = m->grids[(size_t)k]->mat_first_use;
            if (a.size() != nv) return fail(VX_ERR_HIP, "ranks disagree about the mesh's materials");
            for (size_t v = 0; v < nv; ++v)
                if (a[v] >= 0 && (fu[v] < 0 || a[v] < fu[v])) fu[v] = a[v];
        }
        for (int k = 0; k < nd; ++k) {
            VX_TRY(finish_materials(m->grids[(size_t)k], fu.data(), fu.size()));
            shard_ids[(size_t)k] = m->grids[(size_t)k]->mat_count;
        }
    }
    // ---- the complete masks: counts, word prefix and traversal structure; setVoxel calls of all shards add up
    uint64_t calls = 0, total_ids = 0;
    for (int k = 0; k < nd; ++k) {
        VX_TRY(sync_counts(m->grids[(size_t)k]));
        calls += m->grids[(size_t)k]->set_calls;
        total_ids += shard_ids[(size_t)k];
    }
    const uint64_t ntri_all = m->meshes[0]->nt;
    for (int d = 0; d < ndst; ++d) {
        vx_grid* gd = m->grids[(size_t)d];
        DeviceGuard dg(gd->device);
        if (want_mat && nd > 1) {
            // gather the shards' ids in shard order (a destination's own ids move to their place first: a copy within the device)
            DevBuf all;
            all.dev = gd->device;
            all.stream = gd->stream;
            VX_HIP(all.ensure((size_t)total_ids * 2 + 16));
            uint64_t off = 0;
            for (int k = 0; k < nd; ++k) {
                vx_grid* gk = m->grids[(size_t)k];
                if (shard_ids[(size_t)k]) {
                    if (gk->device == gd->device) VX_HIP(hipMemcpyAsync(all.as<int16_t>() + off, gk->matids.p, (size_t)shard_ids[(size_t)k] * 2, hipMemcpyDeviceToDevice, gd->stream));
                    else VX_HIP(hipMemcpyPeerAsync(all.as<int16_t>() + off, gd->device, gk->matids.p, gk->device, (size_t)shard_ids[(size_t)k] * 2, gd->stream));
                }
                off += shard_ids[(size_t)k];
            }
            VX_HIP(hipStreamSynchronize(gd->stream));
            gd->mattmp.release();   // (its first-half scratch is spent; every rank's own ids stay in its matids for the other destinations)
            gd->mattmp = all;       // (kept so that the next build finds a block of this size)
            gd->mat_gathered = true;
            gd->mat_gather_count = total_ids;
        }
        gd->set_calls = calls;
        gd->counts_valid = true;
        gd->triangles = ntri_all;
        VX_TRY(vx_grid_refresh(gd));
    }
    return VX_OK;
}

vx_grid* vx_multi_grid(vx_multi* m, int k) { return (m && k >= 0 && k < m->nd) ? m->grids[(size_t)k] : nullptr; }

vx_grid* vx_multi_release_grid(vx_multi* m, int k)
{
    if (!m || k < 0 || k >= m->nd) return nullptr;
    vx_grid* g = m->grids[(size_t)k];
    m->grids[(size_t)k] = nullptr;
    return g;
}

void vx_multi_free(vx_multi* m)
{
    if (!m) return;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->quit = true;
        m->cv_go.notify_all();
    }
    for (auto& t : m->threads) t.join();
    const int prev_device = g_device;
    for (int k = 0; k < m->nd; ++k) {
        if (m->grids[(size_t)k]) vx_grid_free(m->grids[(size_t)k]);
        if (m->meshes[(size_t)k]) vx_mesh_free(m->meshes[(size_t)k]);
    }
    g_device = prev_device;
    delete m;
}

// one-shot form: create the context, build once, hand the destination grids to the caller
vx_status vx_voxelize_multi(const vx_mesh* mesh, float vs, vx_grid_kind kind, int sat_variant, const int* devices, int nd, int all_gather, vx_grid** out)
{
    if (!out) return fail(VX_ERR_INVALID_ARG, "null argument");
    vx_multi* m = nullptr;
    VX_TRY(vx_multi_create(mesh, devices, nd, kind, &m));
    vx_voxelize_opts o{};
    o.sat_variant = sat_variant;
    const vx_status s = vx_multi_voxelize(m, vs, &o, all_gather);
    if (s != VX_OK) { const std::string e = g_err; vx_multi_free(m); return fail(s, e); }
    const int ndst = all_gather ? nd : 1;
    for (int d = 0; d < ndst; ++d) out[d] = vx_multi_release_grid(m, d);
    vx_multi_free(m);
    return VX_OK;
}

// ---- grid -----------------------------------------------------------------------------------------------------
vx_status vx_grid_create(vx_grid_kind kind, uint64_t x, uint64_t y, uint64_t z, float vs, const float origin[3], void* stream, vx_grid** out)
{
    if (!out) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (kind != VX_GRID_BOOL && kind != VX_GRID_AABBSTRUCT && kind != VX_GRID_VEC) return fail(VX_ERR_INVALID_ARG, "unknown grid kind");
    if (x > vx::kMaxDim || y > vx::kMaxDim || z > vx::kMaxDim || x * y * z > kMaxVoxels) return fail(VX_ERR_CAPACITY, "grid too large");
    VX_TRY(need_device(g_device));
    vx_grid* g = new vx_grid();
    g->kind = kind;
    g->set_dev(g_device);
    g->set_stream((hipStream_t)stream);
    const float zero[3] = {0.f, 0.f, 0.f};
    const uint64_t dim[3] = {x, y, z};
    fill_params(g->g, origin ? origin : zero, vs, dim);
    DeviceGuard dg(g->device);
    const vx_status s = init_grid_storage(g);
    if (s != VX_OK) { g->release_all(); delete g; return s; }
    *out = g;
    return VX_OK;
}

vx_status vx_grid_describe(const vx_grid* gc, vx_grid_desc* d)
{
    if (!gc || !d) return fail(VX_ERR_INVALID_ARG, "null argument");
    vx_grid* g = const_cast<vx_grid*>(gc);
    VX_TRY(sync_counts(g));
    VX_TRY(ensure_prefix(g));
    std::memset(d, 0, sizeof(*d));
    for (int a = 0; a < 3; ++a) {
        d->dim[a] = g->g.dim[a];
        d->origin[a] = g->g.org[a];
        d->bbox_min[a] = g->bbmin[a];
        d->bbox_max[a] = g->bbmax[a];
        d->bbox_center[a] = g->bbc[a];
    }
    d->voxel_size = g->g.vs;
    d->num_words = g->g.nwords;
    d->set_calls = g->set_calls;
    d->occupied = g->occupied;
    d->triangles = g->triangles;
    d->kind = g->kind;
    d->device = g->device;
    return VX_OK;
}

vx_status vx_grid_set_voxel(vx_grid* g, uint64_t x, uint64_t y, uint64_t z)
{
    if (!g) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (x >= g->g.dim[0] || y >= g->g.dim[1] || z >= g->g.dim[2]) return fail(VX_ERR_OUT_OF_BOUNDS, "Index out of bounds");
    DeviceGuard dg(g->device);
    const uint64_t i = x + (uint64_t)g->g.dim[0] * (y + (uint64_t)g->g.dim[1] * z);
    VX_HIP(g->list_resolve());  // (a list emission still to come reads the mask / appends to the list this call changes)
    vx::launch_set_bit(g->words.as<uint32_t>(), i, g->stream);
    if (g->kind == VX_GRID_VEC) {
        // append {c - half, c + half} (voxelgridVecEncoding.cpp:27-36); the float recipe is shared with the kernels
        vx_aabb b;
        vx::cell_aabb(g->g, (uint32_t)x, (uint32_t)y, (uint32_t)z, b.minimum);
        const size_t need = (size_t)(g->vec_count + 1) * sizeof(vx_aabb);
        if (g->vec_in_bound) {  // the list lives in the caller's buffer: appending continues in the grid's own storage
            VX_HIP(g->vec.ensure(need * 2));
            if (g->vec_count) VX_HIP(hipMemcpyAsync(g->vec.p, g->bound, (size_t)g->vec_count * sizeof(vx_aabb), hipMemcpyDeviceToDevice, g->stream));
            g->vec_in_bound = false;
        }
        if (need > g->vec.cap) {
            DevBuf nb;
            nb.dev = g->device;
            nb.stream = g->stream;
            VX_HIP(nb.ensure(need * 2));
            if (g->vec_count) VX_HIP(hipMemcpyAsync(nb.p, g->vec.p, (size_t)g->vec_count * sizeof(vx_aabb), hipMemcpyDeviceToDevice, g->stream));
            VX_HIP(hipStreamSynchronize(g->stream));
            g->vec.release();
            g->vec = nb;
        }
        VX_HIP(hipMemcpyAsync(g->vec.as<vx_aabb>() + g->vec_count, &b, sizeof(b), hipMemcpyHostToDevice, g->stream));
        VX_HIP(hipStreamSynchronize(g->stream));
        g->vec_count++;
    }
    g->host_set_calls++;
    g->set_calls++;
    g->coarse_valid = g->prefix_valid = g->occupied_known = false;
    g->has_materials = false;  // ids are per box of the list: a host-side setVoxel (default material, not recorded) invalidates them
    return VX_OK;
}

vx_status vx_grid_test_voxel(const vx_grid* g, uint64_t x, uint64_t y, uint64_t z, int* occ)
{
    if (!g || !occ) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (x >= g->g.dim[0] || y >= g->g.dim[1] || z >= g->g.dim[2]) return fail(VX_ERR_OUT_OF_BOUNDS, "Index out of bounds");
    DeviceGuard dg(g->device);
    const uint64_t i = x + (uint64_t)g->g.dim[0] * (y + (uint64_t)g->g.dim[1] * z);
    uint32_t w = 0;
    VX_HIP(hipMemcpyAsync(&w, g->words.as<uint32_t>() + (i >> 5), 4, hipMemcpyDeviceToHost, g->stream));
    VX_HIP(hipStreamSynchronize(g->stream));
    *occ = (int)((w >> (i & 31)) & 1u);
    return VX_OK;
}

vx_status vx_grid_coords(const vx_grid* g, uint64_t x, uint64_t y, uint64_t z, float out[3])
{
    if (!g || !out) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (x >= g->g.dim[0] || y >= g->g.dim[1] || z >= g->g.dim[2]) return fail(VX_ERR_OUT_OF_BOUNDS, "Index out of bounds");  // voxelgrid.hpp:93-95
    out[0] = vx::cell_centre(g->g.org[0], g->g.vs, (uint32_t)x);
    out[1] = vx::cell_centre(g->g.org[1], g->g.vs, (uint32_t)y);
    out[2] = vx::cell_centre(g->g.org[2], g->g.vs, (uint32_t)z);
    return VX_OK;
}

uint64_t vx_grid_bytes(const vx_grid* gc)
{
    if (!gc) return 0;
    vx_grid* g = const_cast<vx_grid*>(gc);
    switch (g->kind) {
        case VX_GRID_BOOL: return g->g.nwords * 4;     // m_voxel.size() * sizeof(unsigned)
        case VX_GRID_AABBSTRUCT: return g->g.nvox * 28;  // sizeof(AabbInternal) == 28
        case VX_GRID_VEC: return g->vec_count * 24;      // one Aabb per setVoxel call
    }
    return 0;
}

vx_status vx_grid_bitmask(const vx_grid* g, uint32_t* host_words, uint64_t cap)
{
    if (!g || (!host_words && cap)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (cap < g->g.nwords) return fail(VX_ERR_CAPACITY, "bitmask buffer too small");
    DeviceGuard dg(g->device);
    if (g->g.nwords) VX_HIP(hipMemcpyAsync(host_words, g->words.p, (size_t)g->g.nwords * 4, hipMemcpyDeviceToHost, g->stream));
    VX_HIP(hipStreamSynchronize(g->stream));
    return VX_OK;
}

const uint32_t* vx_grid_bitmask_device(const vx_grid* g) { return g ? g->words.as<uint32_t>() : nullptr; }
uint32_t* vx_grid_bitmask_device_mut(vx_grid* g)
{
    if (!g) return nullptr;
    { DeviceGuard dg(g->device); (void)g->list_resolve(); }  // (an emission still to come reads the mask the caller is about to write)
    g->coarse_valid = g->prefix_valid = g->occupied_known = false;
    return g->words.as<uint32_t>();
}
vx_status vx_grid_refresh(vx_grid* g)
{
    if (!g) return fail(VX_ERR_INVALID_ARG, "null argument");
    { DeviceGuard dg(g->device); VX_HIP(g->list_resolve()); }
    g->coarse_valid = g->prefix_valid = g->occupied_known = false;
    VX_TRY(ensure_prefix(g));
    return ensure_coarse(g);
}

vx_status vx_grid_aabbs_device(const vx_grid* gc, vx_aabb* dev_out, uint64_t cap, uint64_t* count)
{
    if (!gc) return fail(VX_ERR_INVALID_ARG, "null argument");
    vx_grid* g = const_cast<vx_grid*>(gc);
    DeviceGuard dg(g->device);
    if (g->kind == VX_GRID_VEC) {
        if (count) *count = g->vec_count;
        const uint64_t n = cap < g->vec_count ? cap : g->vec_count;
        if (n && dev_out && dev_out != g->vec_ptr()) {  // (a bound buffer already holds the list: nothing to copy)
            VX_HIP(g->list_resolve());
            VX_HIP(hipMemcpyAsync(dev_out, g->vec_ptr(), (size_t)n * sizeof(vx_aabb), hipMemcpyDeviceToDevice, g->stream));
        }
        // (a VX_VOXELIZE_LIST_ASYNC build's list in its bound buffer: the count now, the records after vx_grid_list_wait)
        return VX_OK;
    }
    // queue the prefix scan and the emission back to back, then wait once for the count
    VX_HIP(g->list_resolve());
    bool pending = false;
    VX_TRY(prefix_launch(g, &pending));
    if (cap && dev_out && (pending || g->occupied))
        vx::launch_emit_bool_aabbs(g->words.as<uint32_t>(), g->wprefix.as<uint32_t>(), g->g, dev_out, cap, g->stream, g->sel_valid ? g->wsel.as<uint32_t>() : nullptr);
    {   // (as in vx_grid_aabbs_device_async: the traversal structure of an externally written mask, built while the host waits)
        static const bool eager = !(getenv("VOXHIP_EAGER") && atoi(getenv("VOXHIP_EAGER")) == 0);
        if (eager && pending) VX_TRY(ensure_coarse(g));
    }
    VX_TRY(prefix_finish(g, pending));
    if (count) *count = g->occupied;
    VX_HIP(hipGetLastError());
    return VX_OK;
}

vx_status vx_grid_aabbs_device_async(const vx_grid* gc, vx_aabb* dev_out, uint64_t cap, uint64_t* count)
{
    if (!gc) return fail(VX_ERR_INVALID_ARG, "null argument");
    vx_grid* g = const_cast<vx_grid*>(gc);
    if (g->kind == VX_GRID_VEC) return vx_grid_aabbs_device(gc, dev_out, cap, count);  // (a Vec list is deferred by its build: VX_VOXELIZE_LIST_ASYNC)
    DeviceGuard dg(g->device);
    VX_HIP(g->list_resolve());
    // the word prefix on the grid's stream (the ray batch's primitive ids need it there anyway) and its total = the list's length; the
    // emission itself waits for the next ray batch (trace_common) or for whoever reads the list first (list_resolve)
    bool pending = false;
    VX_TRY(prefix_launch(g, &pending));
    // a mask written from outside (the multi-rank exchange) has no traversal structure yet: queued here, it is built while the host waits
    // for the count instead of after it, in front of the caller's ray batch (VOXHIP_EAGER=0: left to the ray batch)
    static const bool eager = !(getenv("VOXHIP_EAGER") && atoi(getenv("VOXHIP_EAGER")) == 0);
    if (eager && pending) VX_TRY(ensure_coarse(g));
    VX_TRY(prefix_finish(g, pending));
    if (count) *count = g->occupied;
    if (cap && dev_out && g->occupied) {
        g->ld.from_mask = true;
        g->ld.tgt = dev_out;
        g->ld.cap = cap;
        g->list_deferred = true;
    }
    VX_HIP(hipGetLastError());
    return VX_OK;
}

vx_status vx_grid_aabbs(const vx_grid* gc, vx_aabb* host_out, uint64_t cap, uint64_t* count)
{
    if (!gc) return fail(VX_ERR_INVALID_ARG, "null argument");
    vx_grid* g = const_cast<vx_grid*>(gc);
    DeviceGuard dg(g->device);
    uint64_t n = 0;
    if (g->kind == VX_GRID_VEC) n = g->vec_count;
    else { VX_TRY(ensure_prefix(g)); n = g->occupied; }
    if (count) *count = n;
    const uint64_t m = cap < n ? cap : n;
    if (!m || !host_out) return VX_OK;
    if (g->kind == VX_GRID_VEC) {
        VX_HIP(g->list_resolve());
        VX_HIP(hipMemcpyAsync(host_out, g->vec_ptr(), (size_t)m * sizeof(vx_aabb), hipMemcpyDeviceToHost, g->stream));
        VX_HIP(hipStreamSynchronize(g->stream));
        return VX_OK;
    }
    DevBuf tmp;
    tmp.dev = g->device;
    tmp.stream = g->stream;
    VX_HIP(tmp.ensure((size_t)m * sizeof(vx_aabb)));
    vx::launch_emit_bool_aabbs(g->words.as<uint32_t>(), g->wprefix.as<uint32_t>(), g->g, tmp.as<vx_aabb>(), m, g->stream, g->sel_valid ? g->wsel.as<uint32_t>() : nullptr);
    hipError_t e = hipMemcpyAsync(host_out, tmp.p, (size_t)m * sizeof(vx_aabb), hipMemcpyDeviceToHost, g->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    tmp.release();
    VX_HIP(e);
    return VX_OK;
}

vx_status vx_grid_list_wait(vx_grid* g)
{
    if (!g) return fail(VX_ERR_INVALID_ARG, "null argument");
    DeviceGuard dg(g->device);
    VX_HIP(g->list_resolve());
    return VX_OK;
}

vx_status vx_grid_bind_aabbs_device(vx_grid* g, vx_aabb* dev_out, uint64_t capacity)
{
    if (!g) return fail(VX_ERR_INVALID_ARG, "null argument");
    { DeviceGuard dg0(g->device); VX_HIP(g->list_resolve()); }
    if (g->vec_in_bound && g->vec_count) {  // keep the current list reachable: move it into the grid's own storage first
        DeviceGuard dg(g->device);
        VX_HIP(g->vec.ensure((size_t)(g->vec_count + 1) * sizeof(vx_aabb)));
        VX_HIP(hipMemcpyAsync(g->vec.p, g->bound, (size_t)g->vec_count * sizeof(vx_aabb), hipMemcpyDeviceToDevice, g->stream));
        VX_HIP(hipStreamSynchronize(g->stream));
    }
    g->vec_in_bound = false;
    g->bound = capacity ? dev_out : nullptr;
    g->bound_cap = dev_out ? capacity : 0;
    return VX_OK;
}

vx_status vx_grid_materials(const vx_grid* g, vx_material* out, uint64_t cap, uint64_t* count)
{
    if (!g || (!out && cap)) return fail(VX_ERR_INVALID_ARG, "null argument");
    const uint64_t n = g->has_materials ? g->materials.size() : 0;
    if (count) *count = n;
    const uint64_t m = cap < n ? cap : n;
    if (m) std::memcpy(out, g->materials.data(), (size_t)m * sizeof(vx_material));
    return VX_OK;
}

vx_status vx_grid_material_ids(const vx_grid* g, int16_t* out, uint64_t cap, uint64_t* count)
{
    if (!g || (!out && cap)) return fail(VX_ERR_INVALID_ARG, "null argument");
    const uint64_t n = g->has_materials ? (g->mat_gathered ? g->mat_gather_count : g->mat_count) : 0;
    if (count) *count = n;
    const uint64_t m = cap < n ? cap : n;
    if (!m) return VX_OK;
    DeviceGuard dg(g->device);
    VX_HIP(hipMemcpyAsync(out, g->mat_gathered ? g->mattmp.p : g->matids.p, (size_t)m * 2, hipMemcpyDeviceToHost, g->stream));
    VX_HIP(hipStreamSynchronize(g->stream));
    return VX_OK;
}

vx_status vx_grid_material_first_use(const vx_grid* g, int64_t* out, uint64_t cap, uint64_t* count)
{
    if (!g || (!out && cap)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (!g->mat_pending && !g->has_materials) return fail(VX_ERR_INVALID_ARG, "the grid was not built with VX_VOXELIZE_MATERIALS");
    const uint64_t n = g->mat_first_use.size();
    if (count) *count = n;
    const uint64_t m = cap < n ? cap : n;
    for (uint64_t i = 0; i < m; ++i) out[i] = (int64_t)g->mat_first_use[i];
    return VX_OK;
}

vx_status vx_grid_finish_materials(vx_grid* g, const int64_t* first_use_min, uint64_t count)
{
    if (!g || (!first_use_min && count)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (!g->mat_pending) return g->has_materials ? VX_OK : fail(VX_ERR_INVALID_ARG, "the grid was not built with VX_VOXELIZE_MATERIALS");
    std::vector<long long> fu(first_use_min, first_use_min + count);
    return finish_materials(g, fu.data(), fu.size());
}

const int16_t* vx_grid_material_ids_device(const vx_grid* g)
{
    if (!g || !g->has_materials) return nullptr;
    if (g->mat_gathered) return g->mat_gather_count ? g->mattmp.as<int16_t>() : nullptr;
    return g->mat_count ? g->matids.as<int16_t>() : nullptr;
}

void vx_grid_free(vx_grid* g)
{
    if (!g) return;
    { DeviceGuard dg(g->device); (void)g->list_resolve(/*drop_unqueued=*/true); g->side_release(); (void)hipStreamSynchronize(g->stream); }
    g->release_all();
    delete g;
}

// ---- rays -----------------------------------------------------------------------------------------------------
static vx_status trace_common(vx_grid* g, vx::TraceIO io)
{
    VX_TRY(ensure_coarse(g));
    const uint32_t* prefix = nullptr;
    void* idx_tmp = nullptr;
    if (io.prim_out || io.hits || io.normal_out) {
        bool pending = false;
        VX_TRY(prefix_launch(g, &pending));  // the ranks need the prefix array on the stream, not the count on the host
        prefix = g->wprefix.as<uint32_t>();
        VX_HIP(g->idxtmp.ensure(vx::trace_idx_bytes(g->g, io.nrays)));
        idx_tmp = g->idxtmp.p;
        if (!io.t_out) {  // the rank / normal / compaction pass reads t
            VX_HIP(g->ttmp.ensure((size_t)io.nrays * 4 + 8));
            io.t_out = g->ttmp.as<float>();
        }
    }
    vx::TraceMips mips;
    mips.bricks3 = g->bricks.as<unsigned long long>();
    mips.w0 = g->words.as<uint32_t>();
    mips.w1 = g->cwords.as<uint32_t>();
    mips.w2 = g->c2words.as<uint32_t>();
    for (int a = 0; a < 3; ++a) { mips.d1[a] = g->cdim[a]; mips.d2[a] = g->c2dim[a]; }
    if (io.cam) {
        VX_HIP(g->camera.ensure(sizeof(vx::Camera)));
        VX_HIP(hipMemcpyAsync(g->camera.p, io.cam, sizeof(vx::Camera), hipMemcpyHostToDevice, g->stream));
        VX_HIP(hipStreamSynchronize(g->stream));  // the host copy lives on the caller's stack
        io.cam_dev = g->camera.as<vx::Camera>();
    }
    const bool list_beside = g->list_deferred;  // VX_VOXELIZE_LIST_ASYNC: the list's emission goes beside this ray batch
    if (list_beside) VX_HIP(g->list_side_begin());
    static const bool rank16 = !(getenv("VOXHIP_RANK16") && atoi(getenv("VOXHIP_RANK16")) == 0);
    // (the mask is allocated with two spare words, so a rank may read its voxel's whole 16-word group only when that lies inside: nwords % 16 == 0)
    const uint32_t* p16 = (prefix && g->sel_valid && rank16 && (g->g.nwords % 16) == 0) ? g->wp16.as<uint32_t>() : nullptr;
    vx::WalkQueue wq;
    vx::launch_trace(g->g, mips, prefix, io, g->small.as<Small>()->trace_counters, &g->trace_phase, idx_tmp, g->stream, p16, list_beside ? &wq : nullptr);
    if (list_beside) VX_HIP(g->list_side_launch(wq.counter, wq.dry_at));
    VX_HIP(hipGetLastError());
    return VX_OK;
}

vx_status vx_trace_device(const vx_grid* gc, const float* dev_rays, uint64_t nrays, float tmin, float tmax, float* dev_t, uint32_t* dev_prim,
                          vx_hit* dev_hits, uint64_t* dev_nhits)
{
    if (!gc || (nrays && !dev_rays)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (dev_hits && !dev_nhits) return fail(VX_ERR_INVALID_ARG, "dev_hits needs dev_num_hits");
    vx_grid* g = const_cast<vx_grid*>(gc);
    DeviceGuard dg(g->device);
    vx::TraceIO io;
    io.rays = dev_rays; io.nrays = nrays; io.tmin = tmin; io.tmax = tmax;
    io.t_out = dev_t; io.prim_out = dev_prim; io.hits = dev_hits; io.nhits = (unsigned long long*)dev_nhits;
    return trace_common(g, io);
}

vx_status vx_trace_primary_device(const vx_grid* gc, const float vi[16], const float pi[16], uint32_t w, uint32_t h, float tmin, float tmax,
                                  float* dev_t, uint32_t* dev_prim)
{
    if (!gc || !vi || !pi || !dev_t) return fail(VX_ERR_INVALID_ARG, "null argument");
    vx_grid* g = const_cast<vx_grid*>(gc);
    DeviceGuard dg(g->device);
    vx::Camera cam;
    std::memcpy(cam.viewInv, vi, 64);
    std::memcpy(cam.projInv, pi, 64);
    cam.width = w;
    cam.height = h;
    vx::TraceIO io;
    io.cam = &cam; io.nrays = (uint64_t)w * h; io.tmin = tmin; io.tmax = tmax; io.t_out = dev_t; io.prim_out = dev_prim;
    return trace_common(g, io);
}

static vx_status args_to_io(const vx_trace_args* a, vx::Camera* cam, vx::TraceIO* io)
{
    if (!a) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (!a->rays && !(a->view_inverse && a->proj_inverse && a->width && a->height)) return fail(VX_ERR_INVALID_ARG, "no rays and no camera");
    if (a->hits && !a->num_hits) return fail(VX_ERR_INVALID_ARG, "hits needs num_hits");
    io->rays = a->rays;
    io->nrays = a->num_rays;
    if (!a->rays) {
        std::memcpy(cam->viewInv, a->view_inverse, 64);
        std::memcpy(cam->projInv, a->proj_inverse, 64);
        cam->width = a->width;
        cam->height = a->height;
        io->cam = cam;
        io->nrays = (uint64_t)a->width * a->height;
    }
    io->tmin = a->tmin; io->tmax = a->tmax; io->tmax_per_ray = a->tmax_per_ray; io->any_hit = a->any_hit != 0;
    io->t_out = a->t; io->prim_out = a->prim; io->normal_out = a->normal; io->shadowed_out = a->shadowed;
    io->hits = a->hits; io->nhits = (unsigned long long*)a->num_hits;
    if (io->any_hit && (io->prim_out || io->normal_out || io->hits)) return fail(VX_ERR_INVALID_ARG, "any_hit reports only `shadowed` (and an arbitrary accepted t)");
    return VX_OK;
}

vx_status vx_trace_ex_device(const vx_grid* gc, const vx_trace_args* args)
{
    if (!gc) return fail(VX_ERR_INVALID_ARG, "null argument");
    vx_grid* g = const_cast<vx_grid*>(gc);
    DeviceGuard dg(g->device);
    vx::Camera cam{};
    vx::TraceIO io;
    VX_TRY(args_to_io(args, &cam, &io));
    return trace_common(g, io);
}

// host-buffer variant: stages every non-null array through pooled device memory
vx_status vx_trace_ex(const vx_grid* gc, const vx_trace_args* args)
{
    if (!gc) return fail(VX_ERR_INVALID_ARG, "null argument");
    vx_grid* g = const_cast<vx_grid*>(gc);
    DeviceGuard dg(g->device);
    vx::Camera cam{};
    vx::TraceIO io;
    VX_TRY(args_to_io(args, &cam, &io));
    if (args->hits) return fail(VX_ERR_UNSUPPORTED, "the compacted hit list is a device-side output: use vx_trace_ex_device");
    const uint64_t n = io.nrays;
    if (!n) return VX_OK;
    DevBuf dr, dtm, dt, dp, dn, ds;
    for (DevBuf* b : {&dr, &dtm, &dt, &dp, &dn, &ds}) { b->dev = g->device; b->stream = g->stream; }
    auto rel = [&]() { for (DevBuf* b : {&dr, &dtm, &dt, &dp, &dn, &ds}) b->release(); };
    hipError_t e = hipSuccess;
    vx_status st = VX_OK;
    if (io.rays) { e = dr.ensure((size_t)n * 24); if (e == hipSuccess) e = hipMemcpyAsync(dr.p, args->rays, (size_t)n * 24, hipMemcpyHostToDevice, g->stream); io.rays = dr.as<float>(); }
    if (e == hipSuccess && io.tmax_per_ray) { e = dtm.ensure((size_t)n * 4); if (e == hipSuccess) e = hipMemcpyAsync(dtm.p, args->tmax_per_ray, (size_t)n * 4, hipMemcpyHostToDevice, g->stream); io.tmax_per_ray = dtm.as<float>(); }
    if (e == hipSuccess && args->t) { e = dt.ensure((size_t)n * 4); io.t_out = dt.as<float>(); }
    if (e == hipSuccess && args->prim) { e = dp.ensure((size_t)n * 4); io.prim_out = dp.as<uint32_t>(); }
    if (e == hipSuccess && args->normal) { e = dn.ensure((size_t)n * 12); io.normal_out = dn.as<float>(); }
    if (e == hipSuccess && args->shadowed) { e = ds.ensure((size_t)n); io.shadowed_out = ds.as<uint8_t>(); }
    if (e == hipSuccess) st = trace_common(g, io);
    if (e == hipSuccess && st == VX_OK) {
        if (args->t) e = hipMemcpyAsync(args->t, dt.p, (size_t)n * 4, hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess && args->prim) e = hipMemcpyAsync(args->prim, dp.p, (size_t)n * 4, hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess && args->normal) e = hipMemcpyAsync(args->normal, dn.p, (size_t)n * 12, hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess && args->shadowed) e = hipMemcpyAsync(args->shadowed, ds.p, (size_t)n, hipMemcpyDeviceToHost, g->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g->stream);
    }
    rel();
    if (st != VX_OK) return st;
    VX_HIP(e);
    return VX_OK;
}

vx_status vx_trace(const vx_grid* gc, const float* host_rays, uint64_t nrays, float tmin, float tmax, float* host_t, uint32_t* host_prim,
                   uint64_t* num_hits)
{
    if (!gc || (nrays && !host_rays)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (num_hits) *num_hits = 0;
    if (!nrays) return VX_OK;
    std::vector<float> tbuf;
    float* tdst = host_t;
    if (!tdst) { tbuf.resize(nrays); tdst = tbuf.data(); }
    vx_trace_args a{};
    a.rays = host_rays; a.num_rays = nrays; a.tmin = tmin; a.tmax = tmax; a.t = tdst; a.prim = host_prim;
    VX_TRY(vx_trace_ex(gc, &a));
    if (num_hits) {
        uint64_t n = 0;
        for (uint64_t i = 0; i < nrays; ++i) n += tdst[i] > 0.0f;
        *num_hits = n;
    }
    return VX_OK;
}

// ---- octree ---------------------------------------------------------------------------------------------------
vx_status vx_octree_build(const vx_mesh* mesh_c, float vs, uint64_t max_items, void* stream, vx_octree** out)
{
    if (!mesh_c || !out) return fail(VX_ERR_INVALID_ARG, "null argument");
    VX_TRY(check_voxel_size(vs));
    vx_mesh* mesh = const_cast<vx_mesh*>(mesh_c);
    VX_TRY(need_device(mesh->device));
    DeviceGuard dg(mesh->device);
    VX_TRY(mesh_to_device(mesh));
    hipStream_t s = (hipStream_t)stream;
    vx_octree* o = new vx_octree();
    o->device = mesh->device;
    o->stream = s;
    o->vs = vs;
    o->max_items = max_items;
    o->items.dev = o->nodebuf.dev = o->device;
    o->items.stream = o->nodebuf.stream = s;
    DevBuf small, recs, ext, units, ubase, btri, scantmp, umask, bhits, hbase, unsorted, sorttmp, ncount, nbase;
    for (DevBuf* b : {&small, &recs, &ext, &units, &ubase, &btri, &scantmp, &umask, &bhits, &hbase, &unsorted, &sorttmp, &ncount, &nbase}) { b->dev = o->device; b->stream = s; }
    Mail* mail = nullptr;
    auto cleanup = [&]() {
        for (DevBuf* b : {&small, &recs, &ext, &units, &ubase, &btri, &scantmp, &umask, &bhits, &hbase, &unsorted, &sorttmp, &ncount, &nbase}) b->release();
        if (mail) (void)hipHostFree(mail);
        mail = nullptr;
    };
    auto bail = [&](vx_status st) { cleanup(); o->items.release(); o->free_nodes(true); delete o; return st; };
#define OCT_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return bail(fail(VX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__))); } while (0)
#define OCT_TRY(expr) do { vx_status s__ = (expr); if (s__ != VX_OK) return bail(s__); } while (0)
    OCT_HIP(ensure_small(small));
    Small* ds = small.as<Small>();
    OCT_HIP(mail_alloc(&mail));
    Extent ex;
    OCT_TRY(compute_extent(mesh, vs, ds, mail, s, &ex, /*morton_error=*/true));
    for (int a = 0; a < 3; ++a) { o->root_min[a] = ex.mn[a]; o->root_max[a] = ex.mx[a]; o->dim[a] = ex.dim[a]; }
    uint64_t maxDim = ex.dim[0] > ex.dim[1] ? ex.dim[0] : ex.dim[1];
    if (ex.dim[2] > maxDim) maxDim = ex.dim[2];
    if (maxDim == 0) { cleanup(); *out = o; return VX_OK; }  // octTree.hpp:571-574
    o->bits = (uint32_t)std::ceil(std::log2((double)maxDim));  // :577-578
    if (o->bits > 21) return bail(fail(VX_ERR_MORTON_BITS, "We support up to 21 bits per axis (max 2^21 voxels per dimension)!"));
    const float root_ext = vs * (float)(1u << o->bits);  // :592
    for (int a = 0; a < 3; ++a) o->root_max[a] = ex.mn[a] + root_ext;
    const uint32_t ntri = (uint32_t)mesh->nt;
    if (ntri == 0) { cleanup(); *out = o; return VX_OK; }  // :696-699 (returns before buildTree)
    vx::GridParams g;
    fill_params(g, ex.mn, vs, ex.dim);
    uint64_t U = 0;
    OCT_TRY(setup_launch(mesh, g, /*sat a7: octTree.hpp:762*/ 0, 0, ntri, 0, (uint32_t)ex.dim[2], recs, units, ubase, scantmp, mail, s, ext));
    const uint32_t* xw = (ex.dim[0] > 65535 || ex.dim[1] > 65535 || ex.dim[2] > 65535) ? ext.as<uint32_t>() : nullptr;
    OCT_TRY(setup_finish(ntri, ubase, btri, mail, s, &U));
    unsigned long long hits = 0;
    if (U) {
        OCT_HIP(umask.ensure((size_t)(U + 1) * 4));
        const uint64_t nUB = (U + 63) / 64;  // hits per block of 64 units from the voxelizer, scanned: the items' positions (as for the Vec grid)
        OCT_HIP(bhits.ensure((size_t)(nUB + 4) * 4));
        OCT_HIP(hbase.ensure((size_t)(nUB + 4) * 4));
        OCT_HIP(ensure_scan_tmp(scantmp, vx::scan_tmp_bytes(nUB), s));
        vx::launch_voxelize(recs.as<vx::TriRec>(), ubase.as<uint32_t>(), btri.as<uint32_t>(), ntri, g, 0, nullptr, 0, 0, umask.as<uint32_t>(), ds->set_calls, s, xw,
                            bhits.as<uint32_t>());
        vx::launch_scan_u32(bhits.as<uint32_t>(), hbase.as<uint32_t>(), nUB, false, scantmp.p, &mail->hits, s, true, 0, nullptr, next_scan_gen(scantmp, s));
        OCT_HIP(hipStreamSynchronize(s));
        hits = mail->hits & kMailValue;
        if (hits >= 0xFFFFFFFFull) return bail(fail(VX_ERR_CAPACITY, "more than 2^32 octree items"));
    }
    o->nitems = hits;
    if (hits) {
        OCT_HIP(unsorted.ensure((size_t)hits * 8));
        OCT_HIP(o->items.ensure((size_t)hits * 8));
        vx::launch_emit_units(recs.as<vx::TriRec>(), ubase.as<uint32_t>(), btri.as<uint32_t>(), ntri, g, umask.as<uint32_t>(), hbase.as<uint32_t>(),
                              nullptr, unsorted.as<uint64_t>(), s, ~0ull, xw);
        const size_t tb = vx::sort_tmp_bytes(hits);
        OCT_HIP(sorttmp.ensure(tb));
        // octTree.hpp:363; the sort ping-pongs between the two buffers: whichever holds the result becomes the octree's item list
        if (vx::launch_sort_u64(unsorted.as<uint64_t>(), o->items.as<uint64_t>(), hits, o->bits ? (int)(3 * o->bits) : 1, sorttmp.p, tb, s) == 0)
            std::swap(unsorted, o->items);
    }
    // node array (octTree.hpp:319-358, :371)
    if (hits && max_items <= vx::kOctDirectMaxItems) {
        // direct form (vx_octree.hip): nodes starting at every item position -> scan -> start / count / links; one wait for the node count
        const uint32_t ni = (uint32_t)hits;
        OCT_HIP(ncount.ensure((size_t)ni + 16));
        OCT_HIP(nbase.ensure(((size_t)ni + 2) * 4));
        OCT_HIP(ensure_scan_tmp(scantmp, vx::scan_tmp_bytes(ni), s));
        vx::launch_oct_depths(o->items.as<uint64_t>(), ni, o->bits, (uint32_t)max_items, ncount.as<uint8_t>(), s);
        vx::launch_scan_u8(ncount.as<uint8_t>(), nbase.as<uint32_t>(), ni, scantmp.p, &mail->occupied, s, 0, next_scan_gen(scantmp, s));
        OCT_HIP(hipStreamSynchronize(s));
        const unsigned long long nn = mail->occupied & kMailValue;
        if (nn == 0 || nn >= 0xFFFFFFFFull) return bail(fail(VX_ERR_CAPACITY, "more than 2^32 octree nodes"));
        OCT_HIP(o->nodebuf.ensure((size_t)nn * sizeof(vx_octree_node)));
        o->dnodes = o->nodebuf.as<vx_octree_node>();
        OCT_HIP(hipMemsetAsync(o->dnodes, 0xFF, (size_t)nn * sizeof(vx_octree_node), s));  // children = 0xFFFFFFFF (none)
        vx::launch_oct_nodes(o->items.as<uint64_t>(), ni, o->bits, nbase.as<uint32_t>(), o->dnodes, s);
        OCT_HIP(hipStreamSynchronize(s));
        o->nnodes = nn;
    } else {
        // level by level: breadth-first expansion + pre-order renumbering (any max_items; also the empty item list's lone root)
        OCT_HIP(vx::build_octree_nodes(o->items.as<uint64_t>(), (uint32_t)hits, o->bits, max_items, &o->dnodes, &o->nnodes, s));
    }
    cleanup();
#undef OCT_HIP
#undef OCT_TRY
    *out = o;
    return VX_OK;
}

uint64_t vx_octree_num_items(const vx_octree* o) { return o ? o->nitems : 0; }
uint64_t vx_octree_num_nodes(const vx_octree* o) { return o ? o->nnodes : 0; }
uint64_t vx_octree_bytes(const vx_octree* o) { return o ? o->nitems * 8 + o->nnodes * 40 : 0; }

vx_status vx_octree_items(const vx_octree* o, uint64_t* host, uint64_t cap)
{
    if (!o || (!host && cap)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (cap < o->nitems) return fail(VX_ERR_CAPACITY, "item buffer too small");
    if (!o->nitems) return VX_OK;
    DeviceGuard dg(o->device);
    VX_HIP(hipMemcpyAsync(host, o->items.p, (size_t)o->nitems * 8, hipMemcpyDeviceToHost, o->stream));
    VX_HIP(hipStreamSynchronize(o->stream));
    return VX_OK;
}

vx_status vx_octree_nodes(const vx_octree* o, vx_octree_node* host, uint64_t cap)
{
    if (!o || (!host && cap)) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (cap < o->nnodes) return fail(VX_ERR_CAPACITY, "node buffer too small");
    if (!o->nnodes) return VX_OK;
    DeviceGuard dg(o->device);
    VX_HIP(hipMemcpyAsync(host, o->dnodes, (size_t)o->nnodes * sizeof(vx_octree_node), hipMemcpyDeviceToHost, o->stream));
    VX_HIP(hipStreamSynchronize(o->stream));
    return VX_OK;
}

vx_status vx_octree_root_bounds(const vx_octree* o, float mn[3], float mx[3])
{
    if (!o || !mn || !mx) return fail(VX_ERR_INVALID_ARG, "null argument");
    std::memcpy(mn, o->root_min, 12);
    std::memcpy(mx, o->root_max, 12);
    return VX_OK;
}

vx_status vx_octree_aabbs_device(const vx_octree* o, vx_aabb* dev_out, uint64_t cap, uint64_t* count)
{
    if (!o) return fail(VX_ERR_INVALID_ARG, "null argument");
    const uint64_t n = o->nnodes == 0 ? 0 : o->nitems;  // octTree.hpp:505-507
    if (count) *count = n;
    const uint64_t m = cap < n ? cap : n;
    if (!m || !dev_out) return VX_OK;
    DeviceGuard dg(o->device);
    vx::launch_emit_morton_aabbs(o->items.as<uint64_t>(), m, o->root_min, o->vs, dev_out, o->stream);
    VX_HIP(hipGetLastError());
    return VX_OK;
}

vx_status vx_octree_aabbs(const vx_octree* o, vx_aabb* host_out, uint64_t cap, uint64_t* count)
{
    if (!o) return fail(VX_ERR_INVALID_ARG, "null argument");
    const uint64_t n = o->nnodes == 0 ? 0 : o->nitems;
    if (count) *count = n;
    const uint64_t m = cap < n ? cap : n;
    if (!m || !host_out) return VX_OK;
    DeviceGuard dg(o->device);
    DevBuf tmp;
    tmp.dev = o->device;
    tmp.stream = o->stream;
    VX_HIP(tmp.ensure((size_t)m * sizeof(vx_aabb)));
    vx::launch_emit_morton_aabbs(o->items.as<uint64_t>(), m, o->root_min, o->vs, tmp.as<vx_aabb>(), o->stream);
    hipError_t e = hipMemcpyAsync(host_out, tmp.p, (size_t)m * sizeof(vx_aabb), hipMemcpyDeviceToHost, o->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(o->stream);
    tmp.release();
    VX_HIP(e);
    return VX_OK;
}

void vx_octree_free(vx_octree* o)
{
    if (!o) return;
    {   // vx_octree_aabbs_device is asynchronous: nothing of this octree may still be in flight when its memory goes back
        DeviceGuard dg(o->device);
        (void)hipStreamSynchronize(o->stream);
        o->items.release(/*in_flight=*/false);
        o->free_nodes(/*in_flight=*/false);
    }
    delete o;
}

// ---- test aid: the octree's item sort on a host array ----------------------------------------------------------
vx_status vx_sort_u64(uint64_t* host_keys, uint64_t n, int bits)
{
    if (!host_keys && n) return fail(VX_ERR_INVALID_ARG, "null argument");
    if (bits < 1 || bits > 64) return fail(VX_ERR_INVALID_ARG, "bits must be in 1..64");
    if (n >= 0xFFFFFFFFull) return fail(VX_ERR_CAPACITY, "more than 2^32 keys");
    if (!n) return VX_OK;
    VX_TRY(need_device(g_device));
    DeviceGuard dg(g_device);
    DevBuf a, b, tmp;
    for (DevBuf* x : {&a, &b, &tmp}) x->dev = g_device;
    auto rel = [&]() { for (DevBuf* x : {&a, &b, &tmp}) x->release(false); };
    hipError_t e = a.ensure((size_t)n * 8);
    if (e == hipSuccess) e = b.ensure((size_t)n * 8);
    const size_t tb = vx::sort_tmp_bytes(n);
    if (e == hipSuccess) e = tmp.ensure(tb);
    if (e == hipSuccess) e = hipMemcpy(a.p, host_keys, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int where = vx::launch_sort_u64(a.as<uint64_t>(), b.as<uint64_t>(), n, bits, tmp.p, tb, nullptr);
        e = hipStreamSynchronize(nullptr);
        if (e == hipSuccess) e = hipMemcpy(host_keys, where == 0 ? a.p : b.p, (size_t)n * 8, hipMemcpyDeviceToHost);
    }
    rel();
    VX_HIP(e);
    return VX_OK;
}

// ---- per-kernel timing (bench / profiling aid) ----------------------------------------------------------------
vx_status vx_profile_enable(int on)
{
    vx::prof_enable(on != 0);
    return VX_OK;
}
vx_status vx_profile_select(const char* kernel_name)
{
    vx::prof_select(kernel_name);
    return VX_OK;
}
vx_status vx_profile_reset(void)
{
    vx::prof_reset();
    return VX_OK;
}
vx_status vx_profile_read(int slot, char* name, size_t name_capacity, double* total_ms, uint64_t* launches)
{
    if (vx::prof_read(slot, name, name_capacity, total_ms, launches) != 0) return fail(VX_ERR_INVALID_ARG, "no such profile slot");
    return VX_OK;
}

// ---- sharding helpers -----------------------------------------------------------------------------------------
void vx_shard_words(uint64_t num_words, int rank, int world, uint64_t* wb, uint64_t* we, uint64_t* padded)
{
    if (world < 1) world = 1;
    const uint64_t chunk = (num_words + (uint64_t)world - 1) / (uint64_t)world;
    uint64_t b = (uint64_t)rank * chunk;
    if (b > num_words) b = num_words;
    uint64_t e = b + chunk;
    if (e > num_words) e = num_words;
    if (wb) *wb = b;
    if (we) *we = e;
    if (padded) *padded = chunk;
}

void vx_shard_range(uint64_t count, int rank, int world, uint64_t* begin, uint64_t* end)
{
    if (world < 1) world = 1;
    const uint64_t chunk = (count + (uint64_t)world - 1) / (uint64_t)world;
    uint64_t b = (uint64_t)rank * chunk;
    if (b > count) b = count;
    uint64_t e = b + chunk;
    if (e > count) e = count;
    if (begin) *begin = b;
    if (end) *end = e;
}

}  // extern "C"
