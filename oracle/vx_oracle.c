/*
 * vx_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity checker for the HIP kernels.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libvoxhip.so) never links,
 * loads or calls anything in oracle/.
 *
 * PARITY UNPINNED (formal status): the reference ships no tests, fixtures or golden vectors
 * for this path, and its sources cannot be compiled in this image without writing stand-ins
 * for glm, tinyobjloader and <print>, which is not allowed.  The only external anchors are
 * the occupied-voxel counts that SURVEY.md section 8(c) / Appendix A recorded from the
 * unmodified reference (cube +-1 at seven voxel sizes; octree bytes for one case);
 * tests/test_oracle.py checks this file against every one of them.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Float op order follows the reference expression by expression;
 * build with -O2 -ffp-contract=off (x86-64 SSE2: no excess precision, no FMA).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float mn[3]; float mx[3]; } vxo_aabb; /* shaders/host_device.h:117-121 */

typedef struct {
    float bmin[3], bmax[3], center[3];
    uint64_t dim[3]; /* width(x), height(y), depth(z) */
} vxo_grid_info;

/* std::min / std::max / glm::min / glm::max restated literally (NaN/zero-sign behaviour) */
static inline float fmin_std(float a, float b) { return (b < a) ? b : a; }
static inline float fmax_std(float a, float b) { return (a < b) ? b : a; }
static inline int imin_std(int a, int b) { return (b < a) ? b : a; }
static inline int imax_std(int a, int b) { return (a < b) ? b : a; }

static inline v3 v3sub(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
/* glm::dot(vec3): tmp = a*b; return tmp.x + tmp.y + tmp.z */
static inline float v3dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* glm::cross */
static inline v3 v3cross(v3 x, v3 y) {
    v3 r = {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
    return r;
}

/* ---------------------------------------------------------------------------------------
 * a2  computeBboxFromAttrib          VoxelBuilder.hpp:198-224 (dup. octTree.hpp:531-557)
 * a3  grid dims                      VoxelBuilder.hpp:347-349
 * ------------------------------------------------------------------------------------- */
void vxo_grid_info_compute(const float* v, size_t nfloats, float vs, vxo_grid_info* gi)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY};
    float mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i + 2 < nfloats; i += 3) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = fmin_std(mn[a], v[i + a]);
            mx[a] = fmax_std(mx[a], v[i + a]);
        }
    }
    for (int a = 0; a < 3; ++a) {
        gi->bmin[a] = mn[a];
        gi->bmax[a] = mx[a];
        gi->center[a] = (mn[a] + mx[a]) * 0.5f;
        gi->dim[a] = (uint64_t)ceilf((mx[a] - mn[a]) / vs);
    }
}

/* ---------------------------------------------------------------------------------------
 * a7  triBoxOverlap (+axisSeparates, aabbAxisSeparates, planeSeparates)
 *     VoxelBuilder.hpp:73-162 (copy at octTree.hpp:395-484)
 * ------------------------------------------------------------------------------------- */
static int axis_separates(v3 axis, float R, v3 p0, v3 p1, v3 p2)
{
    const float eps = 1e-8f;
    const float ax = fabsf(axis.x) + fabsf(axis.y) + fabsf(axis.z);
    if (ax < eps) return 0;
    const float p0d = v3dot(p0, axis);
    const float p1d = v3dot(p1, axis);
    const float p2d = v3dot(p2, axis);
    const float triMin = fmin_std(p0d, fmin_std(p1d, p2d));
    const float triMax = fmax_std(p0d, fmax_std(p1d, p2d));
    return (triMin > R) || (triMax < -R);
}

static int aabb_axis_separates(v3 h, v3 p0, v3 p1, v3 p2)
{
    const float minx = fmin_std(p0.x, fmin_std(p1.x, p2.x));
    const float maxx = fmax_std(p0.x, fmax_std(p1.x, p2.x));
    if (minx > h.x || maxx < -h.x) return 1;
    const float miny = fmin_std(p0.y, fmin_std(p1.y, p2.y));
    const float maxy = fmax_std(p0.y, fmax_std(p1.y, p2.y));
    if (miny > h.y || maxy < -h.y) return 1;
    const float minz = fmin_std(p0.z, fmin_std(p1.z, p2.z));
    const float maxz = fmax_std(p0.z, fmax_std(p1.z, p2.z));
    if (minz > h.z || maxz < -h.z) return 1;
    return 0;
}

static int plane_separates(v3 n, v3 h, v3 p0)
{
    const float eps = 1e-8f;
    const float anx = fabsf(n.x), any = fabsf(n.y), anz = fabsf(n.z);
    const float len = anx + any + anz;
    if (len < eps) return 0;
    const float r = h.x * anx + h.y * any + h.z * anz;
    const float s = v3dot(n, p0);
    return fabsf(s) > r;
}

static int test_edge_axes(v3 e, v3 h, v3 p0, v3 p1, v3 p2)
{
    v3 Lx = {0.0f, -e.z, e.y};
    float Rx = h.y * fabsf(Lx.y) + h.z * fabsf(Lx.z);
    if (axis_separates(Lx, Rx, p0, p1, p2)) return 1;
    v3 Ly = {e.z, 0.0f, -e.x};
    float Ry = h.x * fabsf(Ly.x) + h.z * fabsf(Ly.z);
    if (axis_separates(Ly, Ry, p0, p1, p2)) return 1;
    v3 Lz = {-e.y, e.x, 0.0f};
    float Rz = h.x * fabsf(Lz.x) + h.y * fabsf(Lz.y);
    if (axis_separates(Lz, Rz, p0, p1, p2)) return 1;
    return 0;
}

int vxo_tri_box_overlap(const float c_[3], const float h_[3], const float v0_[3], const float v1_[3],
                        const float v2_[3])
{
    v3 c = {c_[0], c_[1], c_[2]}, h = {h_[0], h_[1], h_[2]};
    v3 v0 = {v0_[0], v0_[1], v0_[2]}, v1 = {v1_[0], v1_[1], v1_[2]}, v2 = {v2_[0], v2_[1], v2_[2]};
    const v3 p0 = v3sub(v0, c), p1 = v3sub(v1, c), p2 = v3sub(v2, c);
    const v3 e0 = v3sub(p1, p0), e1 = v3sub(p2, p1), e2 = v3sub(p0, p2);
    if (aabb_axis_separates(h, p0, p1, p2)) return 0;
    if (test_edge_axes(e0, h, p0, p1, p2)) return 0;
    if (test_edge_axes(e1, h, p0, p1, p2)) return 0;
    if (test_edge_axes(e2, h, p0, p1, p2)) return 0;
    const v3 n = v3cross(e0, e1);
    if (plane_separates(n, h, p0)) return 0;
    return 1;
}

/* ---------------------------------------------------------------------------------------
 * a8  triBoxOverlapSchwarzSeidel (threaded-path SAT)       VoxelBuilder.hpp:226-335
 * ------------------------------------------------------------------------------------- */
static inline int sep_axis(float a, float b, float c, float ra)
{
    const float mn = fminf(a, fminf(b, c));
    const float mx = fmaxf(a, fmaxf(b, c));
    return (mn > ra) || (mx < -ra);
}

int vxo_tri_box_overlap_ss(const float c_[3], const float h_[3], const float v0_[3], const float v1_[3],
                           const float v2_[3])
{
    v3 c = {c_[0], c_[1], c_[2]}, h = {h_[0], h_[1], h_[2]};
    v3 v0 = {v0_[0], v0_[1], v0_[2]}, v1 = {v1_[0], v1_[1], v1_[2]}, v2 = {v2_[0], v2_[1], v2_[2]};
    const v3 p0 = v3sub(v0, c), p1 = v3sub(v1, c), p2 = v3sub(v2, c);
    const v3 e0 = v3sub(p1, p0), e1 = v3sub(p2, p1), e2 = v3sub(p0, p2);

    float minx = fminf(p0.x, fminf(p1.x, p2.x)), maxx = fmaxf(p0.x, fmaxf(p1.x, p2.x));
    if (minx > h.x || maxx < -h.x) return 0;
    float miny = fminf(p0.y, fminf(p1.y, p2.y)), maxy = fmaxf(p0.y, fmaxf(p1.y, p2.y));
    if (miny > h.y || maxy < -h.y) return 0;
    float minz = fminf(p0.z, fminf(p1.z, p2.z)), maxz = fmaxf(p0.z, fmaxf(p1.z, p2.z));
    if (minz > h.z || maxz < -h.z) return 0;

    const v3 es[3] = {e0, e1, e2};
    for (int k = 0; k < 3; ++k) {
        const v3 e = es[k];
        const v3 ae = {fabsf(e.x), fabsf(e.y), fabsf(e.z)};
        float p0d = -p0.z * e.y + p0.y * e.z;
        float p1d = -p1.z * e.y + p1.y * e.z;
        float p2d = -p2.z * e.y + p2.y * e.z;
        float R = h.y * ae.z + h.z * ae.y;
        if (sep_axis(p0d, p1d, p2d, R)) return 0;
        p0d = p0.x * e.z - p0.z * e.x;
        p1d = p1.x * e.z - p1.z * e.x;
        p2d = p2.x * e.z - p2.z * e.x;
        R = h.x * ae.z + h.z * ae.x;
        if (sep_axis(p0d, p1d, p2d, R)) return 0;
        p0d = -p0.y * e.x + p0.x * e.y;
        p1d = -p1.y * e.x + p1.x * e.y;
        p2d = -p2.y * e.x + p2.x * e.y;
        R = h.x * ae.y + h.y * ae.x;
        if (sep_axis(p0d, p1d, p2d, R)) return 0;
    }
    const v3 n = v3cross(e0, e1);
    const v3 an = {fabsf(n.x), fabsf(n.y), fabsf(n.z)};
    const float r = h.x * an.x + h.y * an.y + h.z * an.z;
    const float s = n.x * p0.x + n.y * p0.y + n.z * p0.z;
    if (fabsf(s) > r) return 0;
    return 1;
}

/* a4  VoxelGrid::getCorrds           voxelgrid.hpp:91-100:  m_org + (posvec + 0.5f) * m_voxelSize */
static inline v3 voxel_centre(const float org[3], float vs, size_t x, size_t y, size_t z)
{
    v3 r = {org[0] + ((float)x + 0.5f) * vs, org[1] + ((float)y + 0.5f) * vs, org[2] + ((float)z + 0.5f) * vs};
    return r;
}

/* a5  loadPos                        VoxelBuilder.hpp:356-362 */
static inline v3 load_pos(const float* verts, int vi_)
{
    const size_t vi = (size_t)vi_;
    v3 r = {verts[3 * vi], verts[3 * vi + 1], verts[3 * vi + 2]};
    return r;
}

/* a6  candidate range                VoxelBuilder.hpp:170-184 (threaded copy :497-511) */
typedef struct { int xs, ys, zs, xe, ye, ze; } cand_range;
static cand_range candidate_range(v3 v0, v3 v1, v3 v2, const float gmin[3], float voxelSize,
                                  const uint64_t dim[3])
{
    v3 tmn = {fmin_std(v0.x, fmin_std(v1.x, v2.x)), fmin_std(v0.y, fmin_std(v1.y, v2.y)),
              fmin_std(v0.z, fmin_std(v1.z, v2.z))};
    v3 tmx = {fmax_std(v0.x, fmax_std(v1.x, v2.x)), fmax_std(v0.y, fmax_std(v1.y, v2.y)),
              fmax_std(v0.z, fmax_std(v1.z, v2.z))};
    cand_range r;
    r.xs = imax_std(0, (int)((tmn.x - gmin[0]) / voxelSize));
    r.ys = imax_std(0, (int)((tmn.y - gmin[1]) / voxelSize));
    r.zs = imax_std(0, (int)((tmn.z - gmin[2]) / voxelSize));
    r.xe = imin_std((int)dim[0], (int)((tmx.x - gmin[0]) / voxelSize) + 2);
    r.ye = imin_std((int)dim[1], (int)((tmx.y - gmin[1]) / voxelSize) + 2);
    r.ze = imin_std((int)dim[2], (int)((tmx.z - gmin[2]) / voxelSize) + 2);
    return r;
}

/* ---------------------------------------------------------------------------------------
 * Hit stream.  Both reference drivers visit shapes in order, triangles in order, then
 * z outer / y / x inner, and call setVoxel(x,y,z) for every overlapping candidate:
 *   a9  serial   VoxelBuilder.hpp:367-420 + computeIntersection :164-196  (SAT a7)
 *   a10 threaded VoxelBuilder.hpp:424-541                                   (SAT a8)
 * The sink receives hits in exactly that order.
 * ------------------------------------------------------------------------------------- */
typedef void (*hit_sink)(void* ctx, uint32_t x, uint32_t y, uint32_t z);

static void voxelize_range(const float* verts, const int32_t* idx, size_t tri_begin, size_t tri_end,
                           const vxo_grid_info* gi, float voxelSize, int sat_ss, int serial_flavour,
                           hit_sink sink, void* ctx)
{
    const float hv = voxelSize * 0.5f; /* VoxelBuilder.hpp:407 / :449-452 */
    const float h[3] = {hv, hv, hv};
    /* serial path derives the size back from the half size (:173); threaded uses it directly (:500) */
    const float vSize = serial_flavour ? (hv * 2.0f) : voxelSize;
    for (size_t t = tri_begin; t < tri_end; ++t) {
        const v3 p0 = load_pos(verts, idx[3 * t]);
        const v3 p1 = load_pos(verts, idx[3 * t + 1]);
        const v3 p2 = load_pos(verts, idx[3 * t + 2]);
        const cand_range r = candidate_range(p0, p1, p2, gi->bmin, vSize, gi->dim);
        const float a0[3] = {p0.x, p0.y, p0.z}, a1[3] = {p1.x, p1.y, p1.z}, a2[3] = {p2.x, p2.y, p2.z};
        for (int z = r.zs; z < r.ze; ++z)
            for (int y = r.ys; y < r.ye; ++y)
                for (int x = r.xs; x < r.xe; ++x) {
                    const v3 c = voxel_centre(gi->bmin, voxelSize, (size_t)x, (size_t)y, (size_t)z);
                    const float cc[3] = {c.x, c.y, c.z};
                    const int hit = sat_ss ? vxo_tri_box_overlap_ss(cc, h, a0, a1, a2)
                                           : vxo_tri_box_overlap(cc, h, a0, a1, a2);
                    if (hit) sink(ctx, (uint32_t)x, (uint32_t)y, (uint32_t)z);
                }
    }
}

/* growable (x,y,z) hit bucket == std::vector<glm::uvec3> threadHits[t], VoxelBuilder.hpp:468 */
typedef struct { uint32_t* d; size_t n, cap; } hitvec;
static void hitvec_push(void* ctx, uint32_t x, uint32_t y, uint32_t z)
{
    hitvec* hv = (hitvec*)ctx;
    if (hv->n + 3 > hv->cap) {
        hv->cap = hv->cap ? hv->cap * 2 : 6144;
        hv->d = (uint32_t*)realloc(hv->d, hv->cap * sizeof(uint32_t));
    }
    hv->d[hv->n++] = x; hv->d[hv->n++] = y; hv->d[hv->n++] = z;
}

typedef struct {
    const float* verts; const int32_t* idx; size_t b, e; const vxo_grid_info* gi; float vs; int sat_ss;
    hitvec hits;
} worker_arg;
static void* worker_main(void* p)
{
    worker_arg* w = (worker_arg*)p;
    voxelize_range(w->verts, w->idx, w->b, w->e, w->gi, w->vs, w->sat_ss, 0, hitvec_push, &w->hits);
    return NULL;
}

/*
 * Run a reference driver and feed every hit, in reference order, to `sink`.
 *   threads == 0 : serial driver a9 (SAT a7 unless sat_override >= 0)
 *   threads >= 1 : threaded driver a10 with that many std::thread-equivalents (SAT a8 unless
 *                  overridden); chunk = ceil(T/threads), buckets merged in thread order (:533-537)
 * sat_override: -1 = driver default, 0 = a7, 1 = a8.
 */
static void run_driver(const float* verts, size_t nverts, const int32_t* idx, size_t ntri, float vs,
                       int threads, int sat_override, const vxo_grid_info* gi, hit_sink sink, void* ctx)
{
    (void)nverts;
    if (threads <= 0) {
        const int ss = sat_override < 0 ? 0 : sat_override;
        voxelize_range(verts, idx, 0, ntri, gi, vs, ss, 1, sink, ctx);
        return;
    }
    if (ntri == 0) return; /* VoxelBuilder.hpp:441-445 */
    const int ss = sat_override < 0 ? 1 : sat_override;
    const size_t chunk = (ntri + (size_t)threads - 1) / (size_t)threads;
    worker_arg* wa = (worker_arg*)calloc((size_t)threads, sizeof(worker_arg));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        const size_t b = (size_t)t * chunk;
        if (b >= ntri) break;
        const size_t e = (b + chunk < ntri) ? b + chunk : ntri;
        wa[t].verts = verts; wa[t].idx = idx; wa[t].b = b; wa[t].e = e; wa[t].gi = gi; wa[t].vs = vs;
        wa[t].sat_ss = ss;
        pthread_create(&th[t], NULL, worker_main, &wa[t]);
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < started; ++t) { /* serial merge, thread order */
        for (size_t i = 0; i < wa[t].hits.n; i += 3) sink(ctx, wa[t].hits.d[i], wa[t].hits.d[i + 1], wa[t].hits.d[i + 2]);
        free(wa[t].hits.d);
    }
    free(wa); free(th);
}

/* ---------------------------------------------------------------------------------------
 * per-voxel material ids -- the plumbing the reference keeps commented out, restated as written:
 *   VoxelBuilder.hpp:375-395   materialId = mesh.material_ids[face] (or -1) -> MaterialObj (default or a copy of m_materials[id])
 *   setVoxel -> addMatrialIfNeeded(idx, material)        voxelgridBool.cpp:64, voxelgridAABBstruct.cpp:31  (idx = voxel index)
 *               addMatrialIfNeeded(m_voxelSet, material) voxelgridVecEncoding.cpp:27                       (idx = call number)
 *   addMatrialIfNeeded         voxelgrid.hpp:102-114     known material -> its index, else the next index; m_matIdx[idx] = index
 *   getMatIdx                  voxelgrid.hpp:79-89       the entries >= 0 in index order
 * Materials are passed as VALUE ids (the caller has already identified equal MaterialObj by operator==, obj_loader.h:45-51).
 * Serial driver order (a9).  per_call = 0: Bool / AABBstruct (one id per occupied voxel); 1: Vec (one id per call).
 * out_ids receives getMatIdx(); out_values receives, per index, the value id (== getMatrials() as value ids).
 * Returns the number of ids; *nvalues_used the number of materials.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int16_t* mat_idx; uint64_t n_idx; uint64_t X, Y; int per_call; uint64_t calls;
    int16_t* value_index; int32_t* values_in_order; int32_t nused; int32_t cur_value;
} mat_ctx;
static void mat_set(void* c, uint32_t x, uint32_t y, uint32_t z)
{
    mat_ctx* m = (mat_ctx*)c;
    const uint64_t idx = m->per_call ? m->calls : (x + m->X * (y + m->Y * (uint64_t)z));
    m->calls++;
    if (m->value_index[m->cur_value] < 0) {                  /* voxelgrid.hpp:108-112 */
        m->value_index[m->cur_value] = (int16_t)m->nused;
        m->values_in_order[m->nused++] = m->cur_value;
    }
    if (idx < m->n_idx) m->mat_idx[idx] = m->value_index[m->cur_value];  /* :106 / :113 (the Vec flavour's m_matIdx has X*Y*Z entries) */
}
uint64_t vxo_build_material_ids(const float* verts, size_t nverts, const int32_t* idx, size_t ntri, float vs, int sat_override,
                                const int32_t* tri_value, int32_t nvalues, int per_call, uint64_t ncalls_hint, int16_t* out_ids, uint64_t cap,
                                int32_t* out_values, int32_t* nvalues_used)
{
    vxo_grid_info gi;
    vxo_grid_info_compute(verts, nverts * 3, vs, &gi);
    mat_ctx m;
    memset(&m, 0, sizeof(m));
    m.X = gi.dim[0]; m.Y = gi.dim[1];
    m.per_call = per_call;
    m.n_idx = per_call ? ncalls_hint : gi.dim[0] * gi.dim[1] * gi.dim[2];
    m.mat_idx = (int16_t*)malloc((size_t)(m.n_idx ? m.n_idx : 1) * sizeof(int16_t));
    for (uint64_t i = 0; i < m.n_idx; ++i) m.mat_idx[i] = -1;       /* voxelgrid.hpp:59 */
    m.value_index = (int16_t*)malloc((size_t)(nvalues > 0 ? nvalues : 1) * sizeof(int16_t));
    for (int32_t i = 0; i < nvalues; ++i) m.value_index[i] = -1;
    m.values_in_order = out_values;
    const int ss = sat_override < 0 ? 0 : sat_override;
    for (size_t t = 0; t < ntri; ++t) {
        m.cur_value = tri_value[t];
        voxelize_range(verts, idx, t, t + 1, &gi, vs, ss, 1, mat_set, &m);
    }
    uint64_t n = 0;
    for (uint64_t i = 0; i < m.n_idx; ++i)
        if (m.mat_idx[i] >= 0) { if (n < cap) out_ids[n] = m.mat_idx[i]; ++n; }
    if (nvalues_used) *nvalues_used = m.nused;
    free(m.mat_idx); free(m.value_index);
    return n;
}

/* ---------------------------------------------------------------------------------------
 * a11 VoxelGridBool                  voxelgridBool.cpp:11-16 (ctor), :54-68 (setVoxel), :18-52 (getAabbs)
 * ------------------------------------------------------------------------------------- */
typedef struct { uint32_t* words; uint64_t X, Y; uint64_t set_calls; } bool_ctx;
static void bool_set(void* c, uint32_t x, uint32_t y, uint32_t z)
{
    bool_ctx* b = (bool_ctx*)c;
    const uint64_t i = x + b->X * (y + b->Y * (uint64_t)z); /* map3dto1d voxelgrid.hpp:37-40 */
    b->words[i / 32] |= (1u << (i % 32));
    b->set_calls++; /* m_voxelSet++ counts calls, not unique voxels */
}

uint64_t vxo_bool_num_words(const vxo_grid_info* gi)
{
    const uint64_t n = gi->dim[0] * gi->dim[1] * gi->dim[2];
    return (n + 31) / 32;
}

/* Builds the bitmask; `words` must hold vxo_bool_num_words() zeroed-or-not uint32 (it is zeroed here).
 * Returns the number of setVoxel calls (== m_voxelSet). */
uint64_t vxo_build_bool(const float* verts, size_t nverts, const int32_t* idx, size_t ntri, float vs,
                        int threads, int sat_override, uint32_t* words)
{
    vxo_grid_info gi;
    vxo_grid_info_compute(verts, nverts * 3, vs, &gi);
    memset(words, 0, (size_t)vxo_bool_num_words(&gi) * 4);
    bool_ctx b = {words, gi.dim[0], gi.dim[1], 0};
    run_driver(verts, nverts, idx, ntri, vs, threads, sat_override, &gi, bool_set, &b);
    return b.set_calls;
}

/* map1dto3d (voxelgrid.hpp:42-49) returns a glm::vec3, i.e. the size_t coords converted to float */
static inline void aabb_from_coords(const float org[3], float vs, float gx, float gy, float gz, vxo_aabb* out)
{
    const float half = 0.5f * vs;
    const float c[3] = {org[0] + (gx + 0.5f) * vs, org[1] + (gy + 0.5f) * vs, org[2] + (gz + 0.5f) * vs};
    for (int a = 0; a < 3; ++a) { out->mn[a] = c[a] - half; out->mx[a] = c[a] + half; }
}

/* VoxelGridBool::getAabbs: ascending word, ascending bit (countr_zero).  Returns the count; writes at
 * most `cap` entries (pass cap=0,out=NULL to size). */
uint64_t vxo_bool_aabbs(const uint32_t* words, const uint64_t dim[3], float vs, const float org[3],
                        vxo_aabb* out, uint64_t cap)
{
    const uint64_t total = dim[0] * dim[1] * dim[2];
    const uint64_t nw = (total + 31) / 32;
    uint64_t n = 0;
    for (uint64_t w = 0; w < nw; ++w) {
        uint32_t v = words[w];
        while (v != 0) {
            const int tz = __builtin_ctz(v);
            const uint64_t i = w * 32 + (uint64_t)tz;
            if (i < total) {
                const uint64_t x = i % dim[0], y = (i / dim[0]) % dim[1], z = i / (dim[0] * dim[1]);
                if (n < cap) aabb_from_coords(org, vs, (float)x, (float)y, (float)z, &out[n]);
                ++n;
            } else
                break;
            v &= ~(1u << tz);
        }
    }
    return n;
}

/* ---------------------------------------------------------------------------------------
 * a12 VoxelGridAABBstruct            voxelgridAABBstruct.cpp:5-48, .hpp:7-12
 *     dense {vec3 min, vec3 max, bool isUsed} (28 B); getAabbs filters isUsed in index order.
 * ------------------------------------------------------------------------------------- */
typedef struct { float mn[3]; float mx[3]; uint8_t used; uint8_t pad[3]; } aabb_internal; /* 28 B */
typedef struct { aabb_internal* cells; uint64_t X, Y; float vs; const float* org; } as_ctx;
static void as_set(void* c, uint32_t x, uint32_t y, uint32_t z)
{
    as_ctx* a = (as_ctx*)c;
    const uint64_t i = x + a->X * (y + a->Y * (uint64_t)z);
    vxo_aabb b;
    aabb_from_coords(a->org, a->vs, (float)(size_t)x, (float)(size_t)y, (float)(size_t)z, &b);
    memcpy(a->cells[i].mn, b.mn, 12); memcpy(a->cells[i].mx, b.mx, 12);
    a->cells[i].used = 1;
}

/* Returns count of AABBs; *bytes_out = getMemoryUsageBytes() = 28 * X*Y*Z (voxelgrid.hpp:115-122). */
uint64_t vxo_build_aabbstruct(const float* verts, size_t nverts, const int32_t* idx, size_t ntri, float vs,
                              int threads, int sat_override, vxo_aabb* out, uint64_t cap, uint64_t* bytes_out)
{
    vxo_grid_info gi;
    vxo_grid_info_compute(verts, nverts * 3, vs, &gi);
    const uint64_t total = gi.dim[0] * gi.dim[1] * gi.dim[2];
    aabb_internal* cells = (aabb_internal*)calloc(total ? total : 1, sizeof(aabb_internal));
    as_ctx a = {cells, gi.dim[0], gi.dim[1], vs, gi.bmin};
    run_driver(verts, nverts, idx, ntri, vs, threads, sat_override, &gi, as_set, &a);
    uint64_t n = 0;
    for (uint64_t i = 0; i < total; ++i)
        if (cells[i].used) {
            if (n < cap) { memcpy(out[n].mn, cells[i].mn, 12); memcpy(out[n].mx, cells[i].mx, 12); }
            ++n;
        }
    free(cells);
    if (bytes_out) *bytes_out = total * sizeof(aabb_internal);
    return n;
}

/* ---------------------------------------------------------------------------------------
 * a13 VoxelGridVec                   voxelgridVecEncoding.cpp:10-39: every setVoxel appends, duplicates kept
 * ------------------------------------------------------------------------------------- */
typedef struct { vxo_aabb* out; uint64_t cap, n; float vs; const float* org; } vec_ctx;
static void vec_set(void* c, uint32_t x, uint32_t y, uint32_t z)
{
    vec_ctx* v = (vec_ctx*)c;
    if (v->n < v->cap) aabb_from_coords(v->org, v->vs, (float)(size_t)x, (float)(size_t)y, (float)(size_t)z, &v->out[v->n]);
    v->n++;
}
uint64_t vxo_build_vec(const float* verts, size_t nverts, const int32_t* idx, size_t ntri, float vs,
                       int threads, int sat_override, vxo_aabb* out, uint64_t cap)
{
    vxo_grid_info gi;
    vxo_grid_info_compute(verts, nverts * 3, vs, &gi);
    vec_ctx v = {out, cap, 0, vs, gi.bmin};
    run_driver(verts, nverts, idx, ntri, vs, threads, sat_override, &gi, vec_set, &v);
    return v.n;
}

/* Raw ordered hit list (x,y,z triplets), the common intermediate of every grid flavour. */
uint64_t vxo_hits(const float* verts, size_t nverts, const int32_t* idx, size_t ntri, float vs,
                  int threads, int sat_override, uint32_t* xyz, uint64_t cap_hits)
{
    vxo_grid_info gi;
    vxo_grid_info_compute(verts, nverts * 3, vs, &gi);
    hitvec hv = {NULL, 0, 0};
    run_driver(verts, nverts, idx, ntri, vs, threads, sat_override, &gi, hitvec_push, &hv);
    const uint64_t n = hv.n / 3;
    const uint64_t m = n < cap_hits ? n : cap_hits;
    if (m) memcpy(xyz, hv.d, (size_t)m * 12);
    free(hv.d);
    return n;
}

/* ---------------------------------------------------------------------------------------
 * a16-a19 Octree                     octTree.hpp
 * ------------------------------------------------------------------------------------- */
static uint64_t lut_x[256], lut_y[256], lut_z[256];
static int lut_ready = 0;
static void build_lut(void)
{
    /* octTree.hpp:22-127 tables: morton256_x[i] spreads the 8 bits of i to every 3rd bit; _y = _x<<1; _z = _x<<2 */
    for (uint32_t i = 0; i < 256; ++i) {
        uint64_t s = 0;
        for (int b = 0; b < 8; ++b) s |= (uint64_t)((i >> b) & 1u) << (3 * b);
        lut_x[i] = s; lut_y[i] = s << 1; lut_z[i] = s << 2;
    }
    lut_ready = 1;
}

/* octTree.hpp:211-218, shifts restated verbatim (<<48 then <<24: the top byte falls off the word) */
uint64_t vxo_morton3d(uint32_t x, uint32_t y, uint32_t z)
{
    if (!lut_ready) build_lut();
    uint64_t m = 0;
    m = lut_z[(z >> 16) & 0xFF] | lut_y[(y >> 16) & 0xFF] | lut_x[(x >> 16) & 0xFF];
    m = m << 48 | lut_z[(z >> 8) & 0xFF] | lut_y[(y >> 8) & 0xFF] | lut_x[(x >> 8) & 0xFF];
    m = m << 24 | lut_z[(z)&0xFF] | lut_y[(y)&0xFF] | lut_x[(x)&0xFF];
    return m;
}

/* octTree.hpp:220-229 */
static uint32_t compact_bits(uint64_t v)
{
    v &= 0x1249249249249249ULL;
    v = (v ^ (v >> 2)) & 0x10c30c30c30c30c3ULL;
    v = (v ^ (v >> 4)) & 0x100f00f00f00f00fULL;
    v = (v ^ (v >> 8)) & 0x1f0000ff0000ffULL;
    v = (v ^ (v >> 16)) & 0x1f00000000ffffULL;
    v = (v ^ (v >> 32)) & 0x1fffffULL;
    return (uint32_t)v;
}

typedef struct { uint32_t children[8]; uint32_t start, count; } oct_node; /* 40 B, octTree.hpp:251-277 */

typedef struct {
    uint64_t* items; uint64_t nitems, cap_items;
    oct_node* nodes; uint64_t nnodes, cap_nodes;
    float root_min[3], root_max[3];
    uint64_t max_items; uint64_t max_depth; uint32_t bits;
    float vs;
    uint64_t dim[3];
} vxo_octree;

static void oct_sink(void* c, uint32_t x, uint32_t y, uint32_t z)
{
    vxo_octree* o = (vxo_octree*)c;
    if (o->nitems == o->cap_items) {
        o->cap_items = o->cap_items ? o->cap_items * 2 : 1024;
        o->items = (uint64_t*)realloc(o->items, o->cap_items * 8);
    }
    o->items[o->nitems++] = vxo_morton3d(x, y, z);
}

static int cmp_u64(const void* a, const void* b)
{
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x > y) - (x < y);
}

/* octTree.hpp:319-358 */
static uint32_t build_node(vxo_octree* o, uint32_t begin, uint32_t end, uint32_t depth)
{
    const uint32_t ni = (uint32_t)o->nnodes;
    if (o->nnodes == o->cap_nodes) {
        o->cap_nodes = o->cap_nodes ? o->cap_nodes * 2 : 64;
        o->nodes = (oct_node*)realloc(o->nodes, o->cap_nodes * sizeof(oct_node));
    }
    o->nnodes++;
    o->nodes[ni].start = begin;
    o->nodes[ni].count = end - begin;
    for (int c = 0; c < 8; ++c) o->nodes[ni].children[c] = 0xFFFFFFFFu;
    if (depth >= o->max_depth || (uint64_t)(end - begin) <= o->max_items) return ni;
    const uint32_t levelShift = 3 * ((uint32_t)o->max_depth - 1 - depth);
    uint32_t cur = begin;
    for (int child = 0; child < 8; ++child) {
        if (cur >= end) break;
        const uint32_t cb = cur;
        while (cur < end) {
            const int oct = (int)((o->items[cur] >> levelShift) & 0x7u);
            if (oct != child) break;
            ++cur;
        }
        if (cb == cur) continue;
        const uint32_t ci = build_node(o, cb, cur, depth + 1);
        o->nodes[ni].children[child] = ci;
    }
    return ni;
}

/* Octree::Octree + buildVoxelGrid      octTree.hpp:487-500, :560-809.  The only live driver is the
 * threaded one with SAT a7 (the serial branch is `if (false)`, :618).  threads<=0 is treated as 1. */
vxo_octree* vxo_octree_build(const float* verts, size_t nverts, const int32_t* idx, size_t ntri, float vs,
                             uint64_t max_items, int threads)
{
    vxo_octree* o = (vxo_octree*)calloc(1, sizeof(vxo_octree));
    o->max_items = max_items; o->vs = vs;
    vxo_grid_info gi;
    vxo_grid_info_compute(verts, nverts * 3, vs, &gi);
    for (int a = 0; a < 3; ++a) { o->root_min[a] = gi.bmin[a]; o->root_max[a] = gi.bmax[a]; o->dim[a] = gi.dim[a]; }
    uint64_t maxDim = gi.dim[0] > gi.dim[1] ? gi.dim[0] : gi.dim[1];
    if (gi.dim[2] > maxDim) maxDim = gi.dim[2];
    if (maxDim == 0) return o; /* :571-574 */
    o->bits = (uint32_t)ceil(log2((double)maxDim)); /* :577-578 */
    if (o->bits > 21) { free(o); return NULL; }  /* :583-585 throws */
    o->max_depth = o->bits;
    const float ext = vs * (float)(1u << o->bits); /* :592 */
    for (int a = 0; a < 3; ++a) o->root_max[a] = gi.bmin[a] + ext;
    if (ntri == 0) return o; /* :696-699: returns before buildTree */
    run_driver(verts, nverts, idx, ntri, vs, threads > 0 ? threads : 1, /*sat a7*/ 0, &gi, oct_sink, o);
    qsort(o->items, (size_t)o->nitems, 8, cmp_u64); /* :363 */
    build_node(o, 0, (uint32_t)o->nitems, 0);        /* :371 */
    return o;
}

/* Octree::buildTree (octTree.hpp:360-372) on an item list that is ALREADY sorted: the node array only.  Test aid for item lists
 * too long for qsort in test time (BASELINE configs[4]: 75M items are sorted by the caller with a vectorised sort; sorting
 * uint64 keys has one answer).  The handle owns a copy of the items; max_depth = bits per axis (:590). */
vxo_octree* vxo_octree_from_sorted_items(const uint64_t* items, uint64_t nitems, uint32_t bits, uint64_t max_items)
{
    if (bits > 21 || nitems >= 0xFFFFFFFFull) return NULL;
    vxo_octree* o = (vxo_octree*)calloc(1, sizeof(vxo_octree));
    o->max_items = max_items; o->bits = bits; o->max_depth = bits;
    o->items = (uint64_t*)malloc((size_t)(nitems ? nitems : 1) * 8);
    memcpy(o->items, items, (size_t)nitems * 8);
    o->nitems = o->cap_items = nitems;
    build_node(o, 0, (uint32_t)nitems, 0); /* :371 */
    return o;
}

uint64_t vxo_octree_num_items(const vxo_octree* o) { return o->nitems; }
uint64_t vxo_octree_num_nodes(const vxo_octree* o) { return o->nnodes; }
/* octTree.hpp:512-523 after shrink_to_fit (:803-804) */
uint64_t vxo_octree_bytes(const vxo_octree* o) { return o->nitems * 8 + o->nnodes * 40; }
void vxo_octree_copy_items(const vxo_octree* o, uint64_t* out) { if (o->nitems) memcpy(out, o->items, (size_t)o->nitems * 8); }
void vxo_octree_copy_nodes(const vxo_octree* o, uint32_t* out) { if (o->nnodes) memcpy(out, o->nodes, (size_t)o->nnodes * 40); }
void vxo_octree_root(const vxo_octree* o, float mn[3], float mx[3]) { memcpy(mn, o->root_min, 12); memcpy(mx, o->root_max, 12); }

static void oct_traverse(const vxo_octree* o, uint32_t ni, vxo_aabb* out, uint64_t cap, uint64_t* n)
{
    const oct_node* nd = &o->nodes[ni];
    int leaf = 1;
    for (int c = 0; c < 8; ++c) if (nd->children[c] != 0xFFFFFFFFu) leaf = 0;
    if (leaf) {
        for (uint32_t i = nd->start; i < nd->start + nd->count; ++i) {
            const uint64_t m = o->items[i];
            const uint32_t ix = compact_bits(m), iy = compact_bits(m >> 1), iz = compact_bits(m >> 2);
            if (*n < cap) {
                /* voxelIndexToCenter :237-240, then pos -/+ (vs*0.5f) :382 */
                const float hs = o->vs * 0.5f;
                const float p[3] = {o->root_min[0] + ((float)ix + 0.5f) * o->vs, o->root_min[1] + ((float)iy + 0.5f) * o->vs,
                                    o->root_min[2] + ((float)iz + 0.5f) * o->vs};
                for (int a = 0; a < 3; ++a) { out[*n].mn[a] = p[a] - hs; out[*n].mx[a] = p[a] + hs; }
            }
            (*n)++;
        }
    } else {
        for (int c = 0; c < 8; ++c) if (nd->children[c] != 0xFFFFFFFFu) oct_traverse(o, nd->children[c], out, cap, n);
    }
}
/* Octree::getAabbs                    octTree.hpp:502-510, :374-392 */
uint64_t vxo_octree_aabbs(const vxo_octree* o, vxo_aabb* out, uint64_t cap)
{
    uint64_t n = 0;
    if (o->nnodes == 0) return 0;
    oct_traverse(o, 0, out, cap, &n);
    return n;
}
void vxo_octree_free(vxo_octree* o) { if (o) { free(o->items); free(o->nodes); free(o); } }

/* ---------------------------------------------------------------------------------------
 * a21 hitAabb                         shaders/raytrace.rint:46-56
 * a22 rint main + Vulkan hit interval shaders/raytrace.rint:58-71, raytrace.rgen:50-51
 *     GLSL min/max on finite operands; `1.0 / dir` restated as IEEE division.
 * ------------------------------------------------------------------------------------- */
float vxo_hit_aabb(const vxo_aabb* b, const float o[3], const float d[3])
{
    float tmn[3], tmx[3];
    for (int a = 0; a < 3; ++a) {
        const float inv = 1.0f / d[a];
        const float tbot = inv * (b->mn[a] - o[a]);
        const float ttop = inv * (b->mx[a] - o[a]);
        tmn[a] = fminf(ttop, tbot);
        tmx[a] = fmaxf(ttop, tbot);
    }
    const float t0 = fmaxf(tmn[0], fmaxf(tmn[1], tmn[2]));
    const float t1 = fminf(tmx[0], fminf(tmx[1], tmx[2]));
    return t1 > fmaxf(t0, 0.0f) ? t0 : -1.0f;
}

/*
 * Brute-force first hit over an AABB list: what traversal of the reference's BLAS must return for
 * opaque geometry.  A candidate is reported iff tHit > 0 (rint:69) and accepted iff
 * tmin <= tHit <= current closest (Vulkan procedural-hit rule); closest wins, ties keep the lower
 * primitive id (the reference leaves ties to the driver).  Miss: t = -1, prim = 0xFFFFFFFF.
 * rays = 6 floats each (origin xyz, direction xyz).
 */
typedef struct { const vxo_aabb* boxes; uint64_t n; const float* rays; uint64_t r0, r1; float tmin, tmax; float* t; uint32_t* prim; } trace_arg;
/* fminf / fmaxf written out (the operand that is not a NaN; the smaller / larger otherwise) so that the brute-force loop does
 * not call into libm 12 times per box; tests/test_oracle.py checks hit_aabb_inv == vxo_hit_aabb, NaN cases included. */
static inline float fmin_(float a, float b) { return (a < b || b != b) ? a : b; }
static inline float fmax_(float a, float b) { return (a > b || b != b) ? a : b; }
/* vxo_hit_aabb with invDir = 1.0 / dir (rint:48) computed once per ray instead of once per box: the same floats */
static inline float hit_aabb_inv(const vxo_aabb* b, const float o[3], const float inv[3])
{
    const float bx = inv[0] * (b->mn[0] - o[0]), tx = inv[0] * (b->mx[0] - o[0]);
    const float by = inv[1] * (b->mn[1] - o[1]), ty = inv[1] * (b->mx[1] - o[1]);
    const float bz = inv[2] * (b->mn[2] - o[2]), tz = inv[2] * (b->mx[2] - o[2]);
    const float t0 = fmax_(fmin_(tx, bx), fmax_(fmin_(ty, by), fmin_(tz, bz)));
    const float t1 = fmin_(fmax_(tx, bx), fmin_(fmax_(ty, by), fmax_(tz, bz)));
    return t1 > fmax_(t0, 0.0f) ? t0 : -1.0f;
}
float vxo_hit_aabb_fast(const vxo_aabb* b, const float o[3], const float d[3])  /* exported for the equivalence test only */
{
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    return hit_aabb_inv(b, o, inv);
}
static void* trace_main(void* p)
{
    trace_arg* a = (trace_arg*)p;
    for (uint64_t r = a->r0; r < a->r1; ++r) {
        const float* o = a->rays + 6 * r;
        const float* d = o + 3;
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        float best = a->tmax; uint32_t bp = 0xFFFFFFFFu; int found = 0;
        for (uint64_t i = 0; i < a->n; ++i) {
            const float t = hit_aabb_inv(&a->boxes[i], o, inv);
            if (t > 0.0f && t >= a->tmin && (found ? t < best : t <= best)) { best = t; bp = (uint32_t)i; found = 1; }
        }
        a->t[r] = found ? best : -1.0f;
        a->prim[r] = bp;
    }
    return NULL;
}
void vxo_trace_brute(const vxo_aabb* boxes, uint64_t n, const float* rays, uint64_t nrays, float tmin, float tmax,
                     int threads, float* t_out, uint32_t* prim_out)
{
    if (threads < 1) threads = 1;
    trace_arg* ta = (trace_arg*)calloc((size_t)threads, sizeof(trace_arg));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    const uint64_t chunk = (nrays + (uint64_t)threads - 1) / (uint64_t)threads;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        const uint64_t b = (uint64_t)t * chunk;
        if (b >= nrays) break;
        trace_arg x = {boxes, n, rays, b, (b + chunk < nrays) ? b + chunk : nrays, tmin, tmax, t_out, prim_out};
        ta[t] = x;
        pthread_create(&th[t], NULL, trace_main, &ta[t]);
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(ta); free(th);
}

/* a23 primary-ray model               shaders/raytrace.rgen:41-47 with the matrices passed in (column-major,
 * glm layout).  Writes 6 floats per pixel, row-major pixels.  normalize = v * inversesqrt(dot(v,v)) in glm;
 * restated with 1/sqrtf. */
static void primary_ray_pixel(const float viewInv[16], const float projInv[16], uint32_t W, uint32_t H, uint32_t px, uint32_t py, float* out)
{
    const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
    const float dx = u * 2.0f - 1.0f, dy = v * 2.0f - 1.0f;
    float tgt[4];
    /* mat4*vec4 in glm's association: (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
    for (int r = 0; r < 4; ++r) tgt[r] = (projInv[0 * 4 + r] * dx + projInv[1 * 4 + r] * dy) + (projInv[2 * 4 + r] * 1.0f + projInv[3 * 4 + r] * 1.0f);
    const float il = 1.0f / sqrtf((tgt[0] * tgt[0] + tgt[1] * tgt[1]) + tgt[2] * tgt[2]);
    const float n[3] = {tgt[0] * il, tgt[1] * il, tgt[2] * il};
    for (int r = 0; r < 3; ++r) {
        out[r] = viewInv[3 * 4 + r];
        out[3 + r] = (viewInv[0 * 4 + r] * n[0] + viewInv[1 * 4 + r] * n[1]) + viewInv[2 * 4 + r] * n[2];
    }
}
void vxo_primary_rays(const float viewInv[16], const float projInv[16], uint32_t W, uint32_t H, float* rays)
{
    for (uint32_t py = 0; py < H; ++py)
        for (uint32_t px = 0; px < W; ++px) primary_ray_pixel(viewInv, projInv, W, H, px, py, rays + 6 * ((size_t)py * W + px));
}
/* the same for selected pixels (index = py*W + px): images too large to materialise as a ray buffer */
void vxo_primary_rays_pixels(const float viewInv[16], const float projInv[16], uint32_t W, uint32_t H, const uint64_t* pixels, uint64_t n, float* rays)
{
    for (uint64_t i = 0; i < n; ++i) primary_ray_pixel(viewInv, projInv, W, H, (uint32_t)(pixels[i] % W), (uint32_t)(pixels[i] / W), rays + 6 * i);
}

/* ---------------------------------------------------------------------------------------
 * the consumers of the hit in raytrace2.rchit.
 *
 * Shadow query            raytrace2.rchit:103-122: traceRayEXT with gl_RayFlagsTerminateOnFirstHitEXT | Opaque |
 *                         SkipClosestHitShader, tMin 0.001, tMax = distance to the light; isShadowed stays true unless the
 *                         miss shader (raytraceShadow.rmiss:25-28) runs, i.e. shadowed <=> SOME box reports an accepted hit.
 * Cube-face normal        raytrace2.rchit:60-73.  GLSL leaves contraction of `o + d*t` to the compiler; restated without FMA.
 * ------------------------------------------------------------------------------------- */
typedef struct { const vxo_aabb* boxes; uint64_t n; const float* rays; uint64_t r0, r1; float tmin, tmax; const float* tmax_per_ray; uint8_t* out; } any_arg;
static void* any_main(void* p)
{
    any_arg* a = (any_arg*)p;
    for (uint64_t r = a->r0; r < a->r1; ++r) {
        const float* o = a->rays + 6 * r;
        const float* d = o + 3;
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        const float tm = a->tmax_per_ray ? a->tmax_per_ray[r] : a->tmax;
        uint8_t s = 0;
        for (uint64_t i = 0; i < a->n && !s; ++i) {
            const float t = hit_aabb_inv(&a->boxes[i], o, inv);   /* == vxo_hit_aabb, see trace_main */
            if (t > 0.0f && t >= a->tmin && t <= tm) s = 1;
        }
        a->out[r] = s;
    }
    return NULL;
}
void vxo_trace_any_brute_mt(const vxo_aabb* boxes, uint64_t n, const float* rays, uint64_t nrays, float tmin, float tmax,
                            const float* tmax_per_ray, int threads, uint8_t* shadowed)
{
    if (threads < 1) threads = 1;
    any_arg* aa = (any_arg*)calloc((size_t)threads, sizeof(any_arg));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    const uint64_t chunk = (nrays + (uint64_t)threads - 1) / (uint64_t)threads;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        const uint64_t b = (uint64_t)t * chunk;
        if (b >= nrays) break;
        any_arg x = {boxes, n, rays, b, (b + chunk < nrays) ? b + chunk : nrays, tmin, tmax, tmax_per_ray, shadowed};
        aa[t] = x;
        pthread_create(&th[t], NULL, any_main, &aa[t]);
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(aa); free(th);
}
void vxo_trace_any_brute(const vxo_aabb* boxes, uint64_t n, const float* rays, uint64_t nrays, float tmin, float tmax,
                         const float* tmax_per_ray, uint8_t* shadowed)
{
    vxo_trace_any_brute_mt(boxes, n, rays, nrays, tmin, tmax, tmax_per_ray, 8, shadowed);
}

static float sign_glsl(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

void vxo_cube_normal(const vxo_aabb* b, const float o[3], const float d[3], float t, float out[3])
{
    float v[3];
    for (int a = 0; a < 3; ++a) {
        const float wp = o[a] + d[a] * t;                      /* :60 */
        const float c = (b->mn[a] + b->mx[a]) * 0.5f;          /* :65 */
        v[a] = wp - c;
    }
    const float il = 1.0f / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); /* normalize */
    const float nx = v[0] * il, ny = v[1] * il, nz = v[2] * il;
    const float ax = fabsf(nx), ay = fabsf(ny), az = fabsf(nz);
    const float maxC = fmaxf(fmaxf(ax, ay), az);               /* :70 */
    out[0] = out[1] = out[2] = 0.0f;
    if (maxC == ax) out[0] = sign_glsl(nx);                    /* :71-73 */
    else if (maxC == ay) out[1] = sign_glsl(ny);
    else out[2] = sign_glsl(nz);
}

/* ---------------------------------------------------------------------------------------
 * the picture.  One pixel = raytrace.rgen:39-67 (primary ray, tMin 0.001, tMax 10000) -> closest voxel box
 * (brute force over the list) -> raytrace2.rchit:53-137 (cube-face normal, point / directional light, material of
 * matIndices[gl_PrimitiveID], Lambert + ambient for illum >= 1, shadow ray with terminate-on-first-hit and tMax = distance to the
 * light, attenuation 0.3 in shadow, specular of wavefront.glsl:32-48 when lit) or raytrace.rmiss:37 (clearColor * 0.8) ->
 * post.frag:33-37 (pow(c, 1/2.2)), stored as 8-bit RGB.  glm / GLSL functions restated in float32: dot = (x*x' + y*y') + z*z',
 * normalize = v * (1 / sqrt(dot(v, v))), reflect(I, N) = I - 2 dot(N, I) N.
 * materials: 20 floats per record in vx_material order (ambient, diffuse, specular, transmittance, emission, shininess, ior,
 * dissolve, illum as int32, texture id as int32); mat_idx NULL = every box uses record 0.
 * ------------------------------------------------------------------------------------- */
static float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
void vxo_shade_image(const vxo_aabb* boxes, uint64_t n, const int16_t* mat_idx, const float* materials, uint64_t nmat, const float viewInv[16],
                     const float projInv[16], uint32_t W, uint32_t H, const float light_pos[3], float light_intensity, int light_type, const float clear[3],
                     int threads, uint8_t* rgb)
{
    const uint64_t npx = (uint64_t)W * H;
    float* rays = (float*)malloc((size_t)npx * 6 * sizeof(float));
    float* t = (float*)malloc((size_t)npx * sizeof(float));
    uint32_t* prim = (uint32_t*)malloc((size_t)npx * sizeof(uint32_t));
    vxo_primary_rays(viewInv, projInv, W, H, rays);
    vxo_trace_brute(boxes, n, rays, npx, 0.001f, 10000.0f, threads, t, prim);   /* rgen:50-51 */
    float* srays = (float*)malloc((size_t)npx * 6 * sizeof(float));
    float* stmax = (float*)malloc((size_t)npx * sizeof(float));
    uint8_t* shadowed = (uint8_t*)calloc((size_t)npx, 1);
    for (uint64_t i = 0; i < npx; ++i) {
        const float* o = rays + 6 * i; const float* d = o + 3;
        const float tt = t[i] > 0.0f ? t[i] : 0.0f;
        const float wp[3] = {o[0] + d[0] * tt, o[1] + d[1] * tt, o[2] + d[2] * tt};
        float L[3]; float dist = 100000.0f;
        if (light_type == 0) {
            const float l[3] = {light_pos[0] - wp[0], light_pos[1] - wp[1], light_pos[2] - wp[2]};
            dist = sqrtf(dot3(l, l));
            const float il = 1.0f / dist;
            L[0] = l[0] * il; L[1] = l[1] * il; L[2] = l[2] * il;
        } else {
            const float il = 1.0f / sqrtf(dot3(light_pos, light_pos));
            L[0] = light_pos[0] * il; L[1] = light_pos[1] * il; L[2] = light_pos[2] * il;
        }
        srays[6 * i + 0] = wp[0]; srays[6 * i + 1] = wp[1]; srays[6 * i + 2] = wp[2];
        srays[6 * i + 3] = L[0]; srays[6 * i + 4] = L[1]; srays[6 * i + 5] = L[2];
        stmax[i] = dist;
    }
    vxo_trace_any_brute_mt(boxes, n, srays, npx, 0.001f, 10000.0f, stmax, threads, shadowed);   /* rchit:103-122 */
    for (uint64_t i = 0; i < npx; ++i) {
        float c[3] = {clear[0] * 0.8f, clear[1] * 0.8f, clear[2] * 0.8f};          /* rmiss:37 */
        if (t[i] > 0.0f) {
            const float* o = rays + 6 * i; const float* d = o + 3;
            const float* L = srays + 6 * i + 3;
            float N[3];
            vxo_cube_normal(&boxes[prim[i]], o, d, t[i], N);                        /* rchit:60-73 */
            const float dist = stmax[i];
            const float li = light_type == 0 ? light_intensity / (dist * dist) : light_intensity;   /* rchit:80-90 */
            const uint64_t mi = mat_idx ? (uint64_t)(mat_idx[prim[i]] < 0 ? 0 : mat_idx[prim[i]]) : 0;
            const float* m = materials + 20 * (mi < nmat ? mi : 0);
            const int illum = ((const int32_t*)m)[18];
            const float dnl = fmaxf(dot3(N, L), 0.0f);                              /* wavefront.glsl:25 */
            float diff[3] = {m[3] * dnl, m[4] * dnl, m[5] * dnl};
            if (illum >= 1) { diff[0] += m[0]; diff[1] += m[1]; diff[2] += m[2]; }
            float att = 0.3f, spec[3] = {0.0f, 0.0f, 0.0f};
            if (dot3(N, L) > 0.0f && !shadowed[i]) {                                /* rchit:99-133 */
                att = 1.0f;
                if (illum >= 2) {                                                   /* wavefront.glsl:32-48 */
                    const float kPi = 3.14159265f, kSh = fmaxf(m[15], 4.0f);
                    const float kE = (2.0f + kSh) / (2.0f * kPi);
                    const float nd[3] = {-d[0], -d[1], -d[2]};
                    const float iv = 1.0f / sqrtf(dot3(nd, nd));
                    const float V[3] = {nd[0] * iv, nd[1] * iv, nd[2] * iv};
                    const float I[3] = {-L[0], -L[1], -L[2]};
                    const float k2 = 2.0f * dot3(N, I);
                    const float R[3] = {I[0] - N[0] * k2, I[1] - N[1] * k2, I[2] - N[2] * k2};
                    const float sp = kE * powf(fmaxf(dot3(V, R), 0.0f), kSh);
                    spec[0] = m[6] * sp; spec[1] = m[7] * sp; spec[2] = m[8] * sp;
                }
            }
            for (int k = 0; k < 3; ++k) c[k] = li * att * (diff[k] + spec[k]);     /* rchit:136 */
        }
        for (int k = 0; k < 3; ++k) {
            const float cl = fminf(fmaxf(c[k], 0.0f), 1.0f);                        /* 8-bit UNORM target */
            rgb[3 * i + k] = (uint8_t)lroundf(powf(cl, 1.0f / 2.2f) * 255.0f);      /* post.frag:36 */
        }
    }
    free(rays); free(t); free(prim); free(srays); free(stmax); free(shadowed);
}
