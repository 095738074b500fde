/*
 * voxhip.h -- C ABI of libvoxhip.so: MI355X (gfx950) implementation of the reference's voxelizer hot path.
 *
 * The reference (MatBayern/Raytracing-Voxilizer-Vulkan-Intresection) has no FFI layer; its boundary for this
 * path is the C++ template API  VoxelBuilder<T>{path}.buildVoxelGrid(vs) -> T,  T::getAabbs(),  Octree{path,vs}
 * and the GLSL intersection shader raytrace.rint.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root).  The C++ facade in
 * raytracing-voxilizer-vulkan-intresection_amd/cpp/ re-creates the reference classes on top of this ABI.
 *
 * Conventions
 *   - plain pointers and sizes only; handles are opaque; every function returns vx_status (VX_OK == 0) unless
 *     stated otherwise; vx_last_error() gives the message of the calling thread's last failure.
 *   - "host" pointers are ordinary memory, "dev" pointers are HIP device memory on the handle's device.
 *   - `stream` arguments are hipStream_t passed as void* (NULL = the default stream).
 *   - handles are not thread-safe; distinct handles may be used from distinct threads.
 *   - all compute runs in HIP kernels; there is no CPU fallback: without a usable device the compute entry
 *     points fail with VX_ERR_NO_DEVICE.
 *
 * Data contracts (identical to the reference)
 *   - occupancy bitmask: uint32 words, LSB first, voxel index i = x + X*(y + Y*z)          (voxelgrid.hpp:37-40,
 *     voxelgridBool.cpp:54-68)
 *   - vx_aabb: 6 x float32, tightly packed, 24 B                                           (shaders/host_device.h:117-121)
 *   - AABB list order: VX_GRID_BOOL / VX_GRID_AABBSTRUCT ascending voxel index, no duplicates;
 *     VX_GRID_VEC triangle-major then z,y,x with duplicates; octree ascending Morton code with duplicates.
 */
#ifndef VOXHIP_H
#define VOXHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vx_status {
    VX_OK = 0,
    VX_ERR_INVALID_ARG = 1,
    VX_ERR_PATH = 2,          /* "Path does not exist!"                  VoxelBuilder.hpp:54-56, octTree.hpp:300-302 */
    VX_ERR_PARSE = 3,         /* "Colud not get valid reader! ..."       VoxelBuilder.hpp:63-65 */
    VX_ERR_OUT_OF_BOUNDS = 4, /* "Index out of bounds"                   voxelgrid.hpp:68-70, voxelgridBool.cpp:57-59 */
    VX_ERR_MORTON_BITS = 5,   /* "We support up to 21 bits per axis ..." octTree.hpp:583-585 */
    VX_ERR_NO_DEVICE = 6,
    VX_ERR_HIP = 7,
    VX_ERR_CAPACITY = 8,      /* caller buffer too small, or an internal 32-bit counter would overflow */
    VX_ERR_UNSUPPORTED = 9
} vx_status;

typedef struct vx_aabb { float minimum[3]; float maximum[3]; } vx_aabb; /* shaders/host_device.h:117-121 */

typedef enum vx_grid_kind {
    VX_GRID_BOOL = 0,       /* VoxelGridBool        voxelgridBool.{hpp,cpp} */
    VX_GRID_AABBSTRUCT = 1, /* VoxelGridAABBstruct  voxelgridAABBstruct.{hpp,cpp} */
    VX_GRID_VEC = 2         /* VoxelGridVec         voxelgridVecEncoding.{hpp,cpp} */
} vx_grid_kind;

/* MaterialObj (common/obj_loader.h:32-52) as VoxelBuilder copies it out of tinyobj::material_t (VoxelBuilder.hpp:383-394) */
typedef struct vx_material {
    float ambient[3], diffuse[3], specular[3], transmittance[3], emission[3];
    float shininess, ior, dissolve;
    int32_t illum;
    int32_t texture_id;
} vx_material;

typedef struct vx_mesh vx_mesh;     /* parsed OBJ: what VoxelBuilder keeps in m_attribs / m_shapes / m_materials */
typedef struct vx_grid vx_grid;     /* a VoxelGrid<T> */
typedef struct vx_octree vx_octree; /* an Octree */

typedef struct vx_grid_desc {
    uint64_t dim[3];     /* m_x, m_y, m_z                                             voxelgrid.hpp:19-21 */
    float voxel_size;    /* m_voxelSize */
    float origin[3];     /* m_org (== bbox min for built grids) */
    float bbox_min[3], bbox_max[3], bbox_center[3]; /* VoxelBuilder.hpp:198-224 (zero for vx_grid_create) */
    uint64_t num_words;  /* ceil(X*Y*Z/32) */
    uint64_t set_calls;  /* m_voxelSet: number of setVoxel calls (duplicates counted)  voxelgridBool.cpp:67 */
    uint64_t occupied;   /* distinct occupied voxels (popcount of the bitmask) */
    uint64_t triangles;  /* "Total triangles processed"                                VoxelBuilder.hpp:417 */
    int32_t kind;        /* vx_grid_kind */
    int32_t device;
} vx_grid_desc;

typedef struct vx_voxelize_opts {
    int32_t sat_variant;  /* 0: triBoxOverlap (serial driver, VoxelBuilder.hpp:118-162, inParaell=false);
                             1: triBoxOverlapSchwarzSeidel (threaded driver, :226-335, inParaell=true) */
    int32_t flags;        /* VX_VOXELIZE_* bits (0 = the reference as it runs today) */
    uint64_t word_begin;  /* multi-GPU shard: only bitmask words [word_begin, word_end) are written by this    */
    uint64_t word_end;    /* call (both 0 = whole grid).  Triangle range for VX_GRID_VEC shards:               */
    uint64_t tri_begin;   /* only triangles [tri_begin, tri_end) are voxelized (both 0 = all).                 */
    uint64_t tri_end;
    void* stream;         /* hipStream_t the grid's kernels run on */
    int32_t shard_rank;   /* word shard BY RANK: with shard_world > 1 (and word_begin == word_end == 0) the build derives            */
    int32_t shard_world;  /* [word_begin, word_end) = vx_shard_words(num_words, shard_rank, shard_world) from its own bounding-box   */
                          /* pass -- the caller need not know the grid's word count beforehand (0 or 1: whole grid)                 */
} vx_voxelize_opts;

/* vx_voxelize_opts.flags */
#define VX_VOXELIZE_MATERIALS 1 /* also fill the per-voxel material ids: setVoxel -> addMatrialIfNeeded, the plumbing the reference keeps
                                   commented out (VoxelBuilder.hpp:375-395, voxelgridBool.cpp:64, voxelgridAABBstruct.cpp:31,
                                   voxelgridVecEncoding.cpp:27).  A word / triangle shard leaves the ids PENDING: the index a material
                                   gets is the order of its first use over the WHOLE build, so the shards' first uses are combined
                                   first (vx_grid_material_first_use -> element-wise minimum over the shards ->
                                   vx_grid_finish_materials); an unsharded build finishes by itself. */
#define VX_VOXELIZE_LIST_ASYNC 2 /* VX_GRID_VEC (ignored with VX_VOXELIZE_MATERIALS): the build returns with the list's LENGTH known but
                                   without having queued the kernel that writes its records.  Nothing the library queues after a build --
                                   traversal structure, word prefix, ray batches -- reads the list, and a ray batch ends with a long drain
                                   in which most of the GPU idles: the next vx_trace* call on the grid queues the emission on a
                                   low-priority side stream of the handle, beside its ray kernel.  vx_grid_aabbs_device on the BOUND
                                   buffer (vx_grid_bind_aabbs_device) then returns the count at once; the records are complete for work
                                   queued on the grid's stream after vx_grid_list_wait(g).  Every other reader of the list in this API
                                   (host copies, copies into another buffer, re-binding, setVoxel, the next build, free) waits by
                                   itself; without a ray batch in between the emission runs on the grid's stream when first asked for. */

/* ---- library ------------------------------------------------------------------------------------------- */
const char* vx_last_error(void);
const char* vx_status_string(vx_status s);
int vx_device_count(void);           /* HIP devices visible (0 when there is none / no driver) */
vx_status vx_set_device(int device); /* device used by handles created afterwards on this thread */
vx_status vx_release_cached_memory(void); /* return the library's pooled device blocks to HIP */

/* ---- mesh: replaces VoxelBuilder::readObjFile (VoxelBuilder.hpp:51-70) / Octree::readObjFile (octTree.hpp:298-316) */
vx_status vx_mesh_load_obj(const char* path, vx_mesh** out);
vx_status vx_mesh_from_arrays(const float* host_xyz, size_t num_vertices, const int32_t* host_tri_indices,
                              size_t num_triangles, vx_mesh** out);
/* borrow arrays already resident in HBM (no copy; they must outlive the mesh) */
vx_status vx_mesh_from_device(const float* dev_xyz, size_t num_vertices, const int32_t* dev_tri_indices,
                              size_t num_triangles, vx_mesh** out);
size_t vx_mesh_num_vertices(const vx_mesh* m);
size_t vx_mesh_num_triangles(const vx_mesh* m);
/* host copies (NULL for vx_mesh_from_device meshes) */
const float* vx_mesh_host_vertices(const vx_mesh* m);
const int32_t* vx_mesh_host_indices(const vx_mesh* m);
/* materials: what tinyobj hands VoxelBuilder as GetMaterials() and shape.mesh.material_ids (VoxelBuilder.hpp:69,375-381):
 * the records of the OBJ's mtllib files in file order and one id per triangle (-1 = none).  vx_mesh_set_materials attaches
 * the same to a mesh built from arrays (ids are copied; pass NULL ids for "no face has a material"). */
size_t vx_mesh_num_materials(const vx_mesh* m);
vx_status vx_mesh_materials(const vx_mesh* m, vx_material* out, size_t capacity);
const int32_t* vx_mesh_host_material_ids(const vx_mesh* m); /* num_triangles entries; NULL when the mesh has no materials */
vx_status vx_mesh_set_materials(vx_mesh* m, const vx_material* materials, size_t num_materials, const int32_t* tri_material_ids);
void vx_mesh_free(vx_mesh* m);

/* ---- voxelize: replaces VoxelBuilder<T,inParaell>::buildVoxelGrid (VoxelBuilder.hpp:338-542) ---------------
 * Limits (the reference has none besides memory): at most 2^21 cells per axis -- the bound the reference's own Octree has
 * (octTree.hpp:583-585); the per-triangle candidate ranges are 16 + 16 bits in the triangle record plus 5 + 5 high bits in an
 * extension word read only by grids with an axis above 65535 cells -- and 2^37 cells in total.  Beyond either vx_voxelize fails with
 * VX_ERR_CAPACITY, vx_octree_build with VX_ERR_MORTON_BITS and the reference's message.  Rays (vx_trace*) work on every grid the
 * builds accept (grids with an axis above 65535 cells are walked by variants of the ray kernel that keep 32-bit cell coordinates). */
vx_status vx_voxelize(const vx_mesh* mesh, float voxel_size, vx_grid_kind kind, const vx_voxelize_opts* opts /*NULL ok*/,
                      vx_grid** out);
/* same, re-using an existing grid handle's device buffers (steady-state loops; no allocation when sizes repeat) */
vx_status vx_voxelize_into(const vx_mesh* mesh, float voxel_size, const vx_voxelize_opts* opts, vx_grid* grid);

/* ---- the same build spread over several GPUs of one process (the call site on a multi-GPU node: hello_vulkan.cpp:677-683).
 * Device k of `devices` voxelizes the bitmask words vx_shard_words(num_words, k, num_devices) of the SAME grid -- contributions are
 * word-disjoint, so their OR is their concatenation -- and the shards travel over xGMI as peer copies (hipMemcpyPeerAsync: one
 * slab per peer link, no collective library needed inside one process; the one-process-per-GPU form of the same exchange is the
 * RCCL all-gather in vx_dist.py).  all_gather = 0: *out_grids receives ONE grid, on devices[0], holding the complete bitmask;
 * all_gather = 1: out_grids[k] receives a complete grid on devices[k] for every k (rays can then be split over the devices).
 * A device may appear more than once in `devices` (logical ranks: rehearsal on a box with fewer GPUs).  VX_GRID_BOOL and
 * VX_GRID_AABBSTRUCT only: VX_GRID_VEC's list order needs triangle shards (vx_voxelize_opts.tri_begin/tri_end). */
vx_status vx_voxelize_multi(const vx_mesh* mesh, float voxel_size, vx_grid_kind kind, int sat_variant, const int* devices, int num_devices, int all_gather,
                            vx_grid** out_grids);
/* The same as a STEADY-STATE entry: vx_multi_create uploads the mesh to every listed device once and starts one worker thread and
 * one grid handle per rank; every vx_multi_voxelize then rebuilds the grid (any voxel size) with no upload, no thread start and --
 * when sizes repeat -- no allocation: each rank derives its word shard from its own bounding-box pass (vx_voxelize_opts.shard_rank /
 * shard_world), the shards are exchanged as peer copies, the destination grids (rank 0, or every rank with all_gather) refresh word
 * prefix and traversal structure.  opts: sat_variant, flags (VX_VOXELIZE_MATERIALS: the shards' first uses are combined and the ids
 * gathered in shard order, see vx_grid_finish_materials) and stream are honoured; the shard fields must be 0.
 * vx_multi_grid: the context's grid of rank k (owned by the context, valid until the next vx_multi_voxelize / vx_multi_free);
 * vx_multi_release_grid hands it to the caller (vx_grid_free), the context then no longer rebuilds that rank. */
typedef struct vx_multi vx_multi;
vx_status vx_multi_create(const vx_mesh* mesh, const int* devices, int num_devices, vx_grid_kind kind, vx_multi** out);
vx_status vx_multi_voxelize(vx_multi* m, float voxel_size, const vx_voxelize_opts* opts /*NULL ok*/, int all_gather);
vx_grid* vx_multi_grid(vx_multi* m, int rank);
vx_grid* vx_multi_release_grid(vx_multi* m, int rank);
void vx_multi_free(vx_multi* m);

/* Materials of a SHARDED build (VX_VOXELIZE_MATERIALS with a word or triangle shard).  addMatrialIfNeeded (voxelgrid.hpp:102-114)
 * numbers materials in the order of their first setVoxel call over the whole build; a shard only sees its own calls.
 * vx_grid_material_first_use: per material VALUE of the mesh (value 0 = MaterialObj{}, then the mesh's records de-duplicated by
 * operator==, in record order) the index of the first triangle that carries it into a setVoxel call of this shard, -1 = none.
 * vx_grid_finish_materials: given the element-wise minimum of those arrays over all shards (-1 = unused everywhere) the grid fills
 * getMatrials() and the ids of ITS voxels / calls (ascending voxel order resp. call order: the shards' id arrays concatenated in
 * shard order are the unsharded build's).  Also callable on an unsharded build (a no-op: it has finished by itself). */
vx_status vx_grid_material_first_use(const vx_grid* g, int64_t* out, uint64_t capacity, uint64_t* count);
vx_status vx_grid_finish_materials(vx_grid* g, const int64_t* first_use_min, uint64_t count);

/* ---- grid: replaces VoxelGrid<T> and its three subclasses ------------------------------------------------ */
/* VoxelGrid ctor (voxelgrid.hpp:52-62): an empty grid of x*y*z voxels */
vx_status vx_grid_create(vx_grid_kind kind, uint64_t x, uint64_t y, uint64_t z, float voxel_size, const float origin[3],
                         void* stream, vx_grid** out);
vx_status vx_grid_describe(const vx_grid* g, vx_grid_desc* out);
/* setVoxel (voxelgridBool.cpp:54-68, voxelgridAABBstruct.cpp:23-47, voxelgridVecEncoding.cpp:19-39) */
vx_status vx_grid_set_voxel(vx_grid* g, uint64_t x, uint64_t y, uint64_t z);
/* occupancy test of one voxel (the reference's VoxelGrid::getVoxel is broken for the Bool grid, voxelgrid.hpp:66-72) */
vx_status vx_grid_test_voxel(const vx_grid* g, uint64_t x, uint64_t y, uint64_t z, int* occupied);
/* getCorrds (voxelgrid.hpp:91-100) */
vx_status vx_grid_coords(const vx_grid* g, uint64_t x, uint64_t y, uint64_t z, float out_xyz[3]);
/* getMemoryUsageBytes (voxelgrid.hpp:115-122): Bool 4*ceil(N/32); AABBstruct 28*N; Vec 24*set_calls */
uint64_t vx_grid_bytes(const vx_grid* g);
/* bitmask access */
vx_status vx_grid_bitmask(const vx_grid* g, uint32_t* host_words, uint64_t capacity_words);
const uint32_t* vx_grid_bitmask_device(const vx_grid* g);
uint32_t* vx_grid_bitmask_device_mut(vx_grid* g); /* for the multi-GPU exchange; call vx_grid_refresh afterwards */
vx_status vx_grid_refresh(vx_grid* g);            /* recount + rebuild derived data after the bitmask was written externally */
/* getAabbs (voxelgridBool.cpp:18-52, voxelgridAABBstruct.cpp:10-22, voxelgridVecEncoding.cpp:15-18).
 * *count receives the list length; at most `capacity` entries are written (capacity 0 = size query). */
vx_status vx_grid_aabbs(const vx_grid* g, vx_aabb* host_out, uint64_t capacity, uint64_t* count);
vx_status vx_grid_aabbs_device(const vx_grid* g, vx_aabb* dev_out, uint64_t capacity, uint64_t* count);
/* VX_GRID_VEC: hand the grid the device buffer its list should be built IN (VoxelGridVec::getAabbs returns a copy of m_voxel,
 * voxelgridVecEncoding.cpp:15-18; a caller that wants the list in its own HBM buffer saves that copy).  Later vx_voxelize_into
 * calls emit straight into dev_out when the list fits `capacity` entries (otherwise into the grid's own storage, as without a
 * binding), and vx_grid_aabbs_device(g, dev_out, ...) then has nothing left to copy.  The buffer must outlive the binding;
 * dev_out NULL / capacity 0 removes it.  No effect on the other flavours (their lists are emitted by vx_grid_aabbs_device). */
vx_status vx_grid_bind_aabbs_device(vx_grid* g, vx_aabb* dev_out, uint64_t capacity);
/* VX_VOXELIZE_LIST_ASYNC builds: makes the grid's stream wait for the list (queues the emission there if no ray batch has taken it
 * along yet).  Does not block the host.  A no-op for every other build. */
vx_status vx_grid_list_wait(vx_grid* g);
/* vx_grid_aabbs_device for the Bool / AABBstruct flavours with the same deferral: the word prefix is queued and *count returned as usual,
 * the kernel that writes the records into dev_out is left to the next vx_trace* call on the grid (low-priority side stream, beside its ray
 * kernel) -- or to vx_grid_list_wait / the next build / any call that changes or reads what it needs, which queue it on the grid's stream.
 * dev_out must stay valid until then.  VX_GRID_VEC: same as vx_grid_aabbs_device. */
vx_status vx_grid_aabbs_device_async(const vx_grid* g, vx_aabb* dev_out, uint64_t capacity, uint64_t* count);
/* getMatrials() / getMatIdx() (voxelgrid.hpp:74-89) of a grid built with VX_VOXELIZE_MATERIALS:
 *   materials     the distinct MaterialObj values in the order addMatrialIfNeeded first met them (equality = MaterialObj::operator==,
 *                 obj_loader.h:45-51: every field except ior and dissolve; a face without usemtl carries MaterialObj{});
 *   material ids  the entries >= 0 of m_matIdx in index order: for VX_GRID_BOOL / VX_GRID_AABBSTRUCT one int16 per occupied voxel in
 *                 ascending voxel order (entry i belongs to box i of vx_grid_aabbs; the value is the material of the LAST setVoxel
 *                 call on that voxel), for VX_GRID_VEC one per setVoxel call in call order (entry i belongs to box i of the list).
 * Without the flag both are empty, as in the reference today.  *count receives the length; capacity 0 = size query. */
vx_status vx_grid_materials(const vx_grid* g, vx_material* host_out, uint64_t capacity, uint64_t* count);
vx_status vx_grid_material_ids(const vx_grid* g, int16_t* host_out, uint64_t capacity, uint64_t* count);
const int16_t* vx_grid_material_ids_device(const vx_grid* g); /* NULL without materials */
void vx_grid_free(vx_grid* g);

/* ---- octree: replaces Octree (octTree.hpp:487-523) ------------------------------------------------------- */
typedef struct vx_octree_node { uint32_t children[8]; uint32_t start; uint32_t count; } vx_octree_node; /* octTree.hpp:251-277 */
vx_status vx_octree_build(const vx_mesh* mesh, float voxel_size, uint64_t max_items_per_leaf, void* stream, vx_octree** out);
uint64_t vx_octree_num_items(const vx_octree* o);
uint64_t vx_octree_num_nodes(const vx_octree* o);
uint64_t vx_octree_bytes(const vx_octree* o); /* getMemoryUsageBytes octTree.hpp:512-523: 8*items + 40*nodes */
vx_status vx_octree_items(const vx_octree* o, uint64_t* host_morton, uint64_t capacity);
vx_status vx_octree_nodes(const vx_octree* o, vx_octree_node* host_nodes, uint64_t capacity);
vx_status vx_octree_root_bounds(const vx_octree* o, float mn[3], float mx[3]);
vx_status vx_octree_aabbs(const vx_octree* o, vx_aabb* host_out, uint64_t capacity, uint64_t* count); /* getAabbs :502-510 */
vx_status vx_octree_aabbs_device(const vx_octree* o, vx_aabb* dev_out, uint64_t capacity, uint64_t* count);
void vx_octree_free(vx_octree* o);

/* ---- rays: replaces the procedural-hit stage  raytrace.rint:46-71 under traceRayEXT (raytrace.rgen:49-64) --
 * For every ray the closest accepted hit over all occupied voxels' AABBs:  t = hitAabb() of that box,
 * accepted iff t > 0 and tmin <= t <= tmax;  prim = index of the box in vx_grid_aabbs() order of the Bool grid
 * (== gl_PrimitiveID); miss: t = -1, prim = 0xFFFFFFFF.  Rays: 6 float32 each (origin xyz, direction xyz).
 * The reference's ray interval is tmin 0.001, tmax 10000 (raytrace.rgen:50-51).                               */
typedef struct vx_hit { uint32_t ray; uint32_t prim; float t; } vx_hit;
vx_status vx_trace(const vx_grid* g, const float* host_rays, uint64_t num_rays, float tmin, float tmax,
                   float* host_t /*NULL ok*/, uint32_t* host_prim /*NULL ok*/, uint64_t* num_hits /*NULL ok*/);
/* device buffers; optional compacted hit list (ballot/prefix compaction) of capacity num_rays; *dev_num_hits is a
 * device uint64 counter the call zeroes first */
vx_status vx_trace_device(const vx_grid* g, const float* dev_rays, uint64_t num_rays, float tmin, float tmax,
                          float* dev_t /*NULL ok*/, uint32_t* dev_prim /*NULL ok*/, vx_hit* dev_hits /*NULL ok*/,
                          uint64_t* dev_num_hits /*NULL ok*/);
/* primary rays generated in-kernel from the reference camera model (raytrace.rgen:41-47): pixel (px,py) of a
 * width x height image, column-major 4x4 viewInverse / projInverse; outputs indexed py*width+px */
vx_status vx_trace_primary_device(const vx_grid* g, const float view_inverse[16], const float proj_inverse[16],
                                  uint32_t width, uint32_t height, float tmin, float tmax, float* dev_t,
                                  uint32_t* dev_prim /*NULL ok*/);

/* Extended query: one struct for every input/output of the ray stage, including the two consumers of the hit in
 * raytrace2.rchit: the cube-face normal (:60-73) and the shadow query (:103-122, gl_RayFlagsTerminateOnFirstHitEXT with
 * tMax = distance to the light).  Pointers are device pointers for vx_trace_ex_device and host pointers for vx_trace_ex;
 * every output is optional. */
typedef struct vx_trace_args {
    const float* rays;           /* 6 f32 per ray; NULL = primary rays from the camera below (raytrace.rgen:41-47) */
    const float* view_inverse;   /* column-major 4x4 (host pointers in both variants) */
    const float* proj_inverse;
    uint32_t width, height;
    uint64_t num_rays;           /* ignored for camera rays (= width*height) */
    float tmin, tmax;
    const float* tmax_per_ray;   /* optional: replaces tmax per ray */
    int32_t any_hit;             /* 1: terminate on the first accepted hit; only `shadowed` (and t) may be requested */
    int32_t reserved;
    float* t;                    /* closest accepted t, -1 on miss (any_hit: some accepted t) */
    uint32_t* prim;              /* gl_PrimitiveID of the closest hit, 0xFFFFFFFF on miss */
    float* normal;               /* 3 f32 per ray: (+-1,0,0)/(0,+-1,0)/(0,0,+-1), zeros on miss */
    uint8_t* shadowed;           /* 1 = an accepted hit exists */
    vx_hit* hits;                /* device variant only: compacted hit list */
    uint64_t* num_hits;
} vx_trace_args;
vx_status vx_trace_ex_device(const vx_grid* g, const vx_trace_args* args);
vx_status vx_trace_ex(const vx_grid* g, const vx_trace_args* args);

/* ---- test aid: the device radix sort the Octree uses for its Morton items (octTree.hpp:363 -> vx_sort.hip), applied to a host array.
 * Keys must have no bit set at or above `bits` (1..64); sorted in place. */
vx_status vx_sort_u64(uint64_t* host_keys, uint64_t n, int bits);

/* ---- measurement aid: per-kernel durations from HIP events recorded on the launch stream (off by default).
 * slot = 0,1,... until VX_ERR_INVALID_ARG; name is the kernel symbol as launched. */
vx_status vx_profile_enable(int on);
/* time only launches of this kernel (bare name, e.g. "k_trace"); NULL or "" = every kernel.  Two event records per
 * launch are not free, so a throughput measurement selects the one kernel it prices. */
vx_status vx_profile_select(const char* kernel_name);
vx_status vx_profile_reset(void);
vx_status vx_profile_read(int slot, char* name, size_t name_capacity, double* total_ms, uint64_t* launches);

/* ---- multi-GPU helpers (host arithmetic only) ----------------------------------------------------------------
 * Word-aligned shard of the bitmask for rank r of n: contributions of different ranks are word-disjoint, so an
 * all-gather of the shards (or a sum/max all-reduce of zero-padded buffers) equals the OR of the full masks. */
void vx_shard_words(uint64_t num_words, int rank, int world, uint64_t* word_begin, uint64_t* word_end,
                    uint64_t* padded_shard_words);
void vx_shard_range(uint64_t count, int rank, int world, uint64_t* begin, uint64_t* end);

#ifdef __cplusplus
}
#endif
#endif /* VOXHIP_H */
