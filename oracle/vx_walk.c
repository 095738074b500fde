/*
 * vx_walk.c -- TEST INFRASTRUCTURE / CPU BASELINE: a grid-walking first-hit tracer on the CPU.
 *
 * The reference has NO CPU ray path: its rays run in the Vulkan ray-tracing pipeline, the driver's BVH hands candidate boxes to
 * raytrace.rint:46-71, and the result per ray is the brute-force minimum that vxo_trace_brute (vx_oracle.c) computes.  This file
 * is (1) the CPU stand-in for the ray stage in bench.py's cpu_baseline (labelled as such), and (2) the scalar statement of the
 * traversal the HIP kernel k_walk implements, so that the enumeration argument below can be tested against the brute force on
 * the CPU (tests/test_oracle.py) before any GPU time is spent.  Only tests/, bench.py's cpu_baseline leg and smoke() may load it.
 *
 * Major-axis slab walk.  Let w be the axis with the largest |direction| component, u and v the other two.  The grid is cut into
 * slabs perpendicular to w at three granularities: 64 cells (blocks), 8 cells (bricks), 1 cell.  For a slab with lattice planes
 * P_near, P_far (in travel order) the ray is within the position tolerance `tol` of the slab for
 *      t in [ta, tb],  ta = inv_w * ((P_near -/+ tol) - o_w),  tb = inv_w * ((P_far +/- tol) - o_w)
 * -- the same expression form as hitAabb's `invDir * (plane - origin)` (rint:49-50), so by monotonicity of float subtraction and
 * multiplication every box of the slab has a COMPUTED entry time >= ta: once the best accepted t is < ta of a slab, no box of
 * that slab or any later one can beat it (exact, no slack).  Inside [ta, tb] the ray's u and v coordinates stay inside
 *      [min(p(ta), p(tb)) - 2 tol, max(p(ta), p(tb)) + 2 tol]
 * (|d_u|, |d_v| <= |d_w|: an error in t moves the point by less than the same error along w, and there are no 1/d_u blow-ups;
 * a zero component simply gives a constant coordinate), so the cells the ray can touch in the slab are inside that rectangle,
 * typically 1x1 .. 2x2 cells of the slab's granularity.  Coarse slabs whose rectangle holds no occupied block / brick are
 * skipped whole; occupied cells in the rectangle of a 1-cell slab go through the exact rint formula on the float box the
 * reference would have built, which is the only arbiter of hit and t.  Result: the brute-force minimum, ties to the lower
 * voxel index (= lower primitive id), bit for bit.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float mn[3], mx[3]; } walk_aabb;

typedef struct {
    const uint32_t* words;  /* reference-layout occupancy bitmask */
    uint32_t* m1;           /* one bit per 8^3 cells */
    uint32_t* m2;           /* one bit per 64^3 cells */
    uint64_t dim[3], d1[3], d2[3];
    float org[3], vs, half;
} walk_grid;

static inline float fmin_(float a, float b) { return (a < b || b != b) ? a : b; }
static inline float fmax_(float a, float b) { return (a > b || b != b) ? a : b; }

static inline int bit_at(const uint32_t* w, uint64_t i) { return (int)((w[i >> 5] >> (i & 31u)) & 1u); }

/* occupancy lookup at level L (0 cells, 1 bricks, 2 blocks), coordinates in cells of that level */
static inline int occ_at(const walk_grid* g, int lvl, const int64_t c[3])
{
    if (lvl == 0) return bit_at(g->words, (uint64_t)c[0] + g->dim[0] * ((uint64_t)c[1] + g->dim[1] * (uint64_t)c[2]));
    if (lvl == 1) return bit_at(g->m1, (uint64_t)c[0] + g->d1[0] * ((uint64_t)c[1] + g->d1[1] * (uint64_t)c[2]));
    return bit_at(g->m2, (uint64_t)c[0] + g->d2[0] * ((uint64_t)c[1] + g->d2[1] * (uint64_t)c[2]));
}

/* the box every grid flavour emits for voxel (x,y,z): c -/+ half with c = org + (i + 0.5) * vs  (voxelgridBool.cpp:37-41) */
static inline void cell_box(const walk_grid* g, const int64_t c[3], walk_aabb* b)
{
    for (int a = 0; a < 3; ++a) {
        const float ctr = g->org[a] + (((float)c[a] + 0.5f) * g->vs);
        b->mn[a] = ctr - g->half;
        b->mx[a] = ctr + g->half;
    }
}

/* hitAabb, rint:46-56, invDir hoisted (the same floats as vxo_hit_aabb) */
static inline float hit_box(const walk_aabb* b, const float o[3], const float inv[3])
{
    const float bx = inv[0] * (b->mn[0] - o[0]), tx = inv[0] * (b->mx[0] - o[0]);
    const float by = inv[1] * (b->mn[1] - o[1]), ty = inv[1] * (b->mx[1] - o[1]);
    const float bz = inv[2] * (b->mn[2] - o[2]), tz = inv[2] * (b->mx[2] - o[2]);
    const float t0 = fmax_(fmin_(tx, bx), fmax_(fmin_(ty, by), fmin_(tz, bz)));
    const float t1 = fmin_(fmax_(tx, bx), fmin_(fmax_(ty, by), fmax_(tz, bz)));
    return t1 > fmax_(t0, 0.0f) ? t0 : -1.0f;
}

typedef struct {
    float o[3], d[3], inv[3];
    int w, u, v;        /* major axis and the other two */
    int sw;             /* +1 / -1: travel direction along w */
    float tol;          /* position tolerance */
    float tn, tf;       /* the ray inside the tolerance-dilated grid box, cut to [0, tmax] */
    float tmin, tmax;
    float best;
    uint64_t best_idx;
    int found;
    uint64_t steps[3];  /* slabs looked at per level (statistics) */
    uint64_t tests;     /* exact slab tests */
} walk_ray;

/* lattice plane number i (in cells) along axis a: the float the DDA uses for it */
static inline float plane(const walk_grid* g, int a, int64_t i) { return g->org[a] + (float)i * g->vs; }

/* [ta, tb] of the slab [i0, i1) (cells) along the major axis, dilated by tol */
static inline void slab_times(const walk_grid* g, const walk_ray* r, int64_t i0, int64_t i1, float* ta, float* tb)
{
    const int w = r->w;
    const float lo = plane(g, w, i0) - r->tol, hi = plane(g, w, i1) + r->tol;
    const float t_lo = r->inv[w] * (lo - r->o[w]), t_hi = r->inv[w] * (hi - r->o[w]);
    *ta = r->sw > 0 ? t_lo : t_hi;
    *tb = r->sw > 0 ? t_hi : t_lo;
}

/* cells (of edge 1<<sh fine cells) along axis a that the ray can touch for t in [ta, tb] */
static inline void minor_range(const walk_grid* g, const walk_ray* r, int a, int sh, uint64_t ncells, float ta, float tb, int64_t* c0, int64_t* c1)
{
    const float pa = r->o[a] + ta * r->d[a], pb = r->o[a] + tb * r->d[a];
    const float tol2 = 2.0f * r->tol;
    const float inv_vs = 1.0f / g->vs;
    const float lo = (fmin_(pa, pb) - tol2 - g->org[a]) * inv_vs, hi = (fmax_(pa, pb) + tol2 - g->org[a]) * inv_vs;
    int64_t i0 = (int64_t)floorf(lo), i1 = (int64_t)floorf(hi);
    if (i0 < 0) i0 = 0;
    if (i1 < 0) i1 = -1;
    i0 >>= sh;
    i1 = i1 < 0 ? -1 : (i1 >> sh);
    if (i1 > (int64_t)ncells - 1) i1 = (int64_t)ncells - 1;
    *c0 = i0;
    *c1 = i1;
}

/* walk the slabs of level lvl (cell edge 1<<(3*lvl)) whose index along w lies in [k_first, k_last] (travel order) */
static void walk_level(const walk_grid* g, walk_ray* r, int lvl, int64_t k_first, int64_t k_last)
{
    const int sh = 3 * lvl, w = r->w, u = r->u, v = r->v;
    const uint64_t* nd = lvl == 0 ? g->dim : (lvl == 1 ? g->d1 : g->d2);
    for (int64_t k = k_first; r->sw > 0 ? k <= k_last : k >= k_last; k += r->sw) {
        int64_t i0 = k << sh, i1 = (k + 1) << sh;
        if (i1 > (int64_t)g->dim[w]) i1 = (int64_t)g->dim[w];
        float ta, tb;
        slab_times(g, r, i0, i1, &ta, &tb);
        if (r->found && r->best < ta) return;    /* no box of this or any later slab can beat the best hit */
        if (ta > r->tf) return;                 /* beyond the ray's interval / the grid */
        if (tb < r->tn) continue;               /* still in front of the interval's start */
        ta = fmax_(ta, r->tn);
        tb = fmin_(tb, r->tf);
        if (!(ta <= tb)) continue;
        r->steps[lvl]++;
        int64_t u0, u1, v0, v1;
        minor_range(g, r, u, sh, nd[u], ta, tb, &u0, &u1);
        minor_range(g, r, v, sh, nd[v], ta, tb, &v0, &v1);
        int any = 0;
        for (int64_t cv = v0; cv <= v1; ++cv)
            for (int64_t cu = u0; cu <= u1; ++cu) {
                int64_t c[3];
                c[w] = k; c[u] = cu; c[v] = cv;
                if (!occ_at(g, lvl, c)) continue;
                if (lvl > 0) { any = 1; continue; }
                walk_aabb b;
                cell_box(g, c, &b);
                const float t = hit_box(&b, r->o, r->inv);
                r->tests++;
                const uint64_t idx = (uint64_t)c[0] + g->dim[0] * ((uint64_t)c[1] + g->dim[1] * (uint64_t)c[2]);
                if (t > 0.0f && t >= r->tmin && t <= r->tmax && (!r->found || t < r->best || (t == r->best && idx < r->best_idx))) {
                    r->best = t; r->best_idx = idx; r->found = 1;
                }
            }
        if (any) {
            /* the eight finer slabs of this one, in travel order */
            const int64_t f0 = k << 3, f1 = (k << 3) + 7;
            const uint64_t* nf = lvl == 2 ? g->d1 : g->dim;
            int64_t a0 = f0, a1 = f1 > (int64_t)nf[w] - 1 ? (int64_t)nf[w] - 1 : f1;
            if (r->sw > 0) walk_level(g, r, lvl - 1, a0, a1); else walk_level(g, r, lvl - 1, a1, a0);
        }
    }
}

static void trace_one(const walk_grid* g, const float* ray, float tmin, float tmax, float* t_out, uint64_t* idx_out, uint64_t stats[4])
{
    walk_ray r;
    memset(&r, 0, sizeof(r));
    for (int a = 0; a < 3; ++a) { r.o[a] = ray[a]; r.d[a] = ray[3 + a]; r.inv[a] = 1.0f / r.d[a]; }   /* rint:48 */
    r.tmin = tmin; r.tmax = tmax;
    *t_out = -1.0f;
    *idx_out = ~0ull;
    const float ax = fabsf(r.d[0]), ay = fabsf(r.d[1]), az = fabsf(r.d[2]);
    if (!(ax > 0.0f || ay > 0.0f || az > 0.0f) || !g->dim[0] || !g->dim[1] || !g->dim[2]) return;  /* degenerate ray (never hits: see vx_oracle.c) / empty grid */
    r.w = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    r.u = (r.w + 1) % 3;
    r.v = (r.w + 2) % 3;
    r.sw = r.d[r.w] > 0.0f ? 1 : -1;
    /* position tolerance 16 * 2^-24 * max |coordinate| of the origin and the grid corners */
    float M = 0.0f, hi[3];
    for (int a = 0; a < 3; ++a) {
        hi[a] = g->org[a] + (float)g->dim[a] * g->vs;
        M = fmax_(M, fmax_(fabsf(r.o[a]), fmax_(fabsf(g->org[a]), fabsf(hi[a]))));
    }
    r.tol = M * 9.5367431640625e-07f;
    /* the ray inside the dilated grid box, cut to [0, tmax] */
    float tn = 0.0f, tf = tmax;
    for (int a = 0; a < 3; ++a) {
        if (r.d[a] == 0.0f) {
            if (r.o[a] < g->org[a] - r.tol || r.o[a] > hi[a] + r.tol) return;
            continue;
        }
        const float t1 = ((g->org[a] - r.tol) - r.o[a]) * r.inv[a], t2 = ((hi[a] + r.tol) - r.o[a]) * r.inv[a];
        tn = fmax_(tn, fmin_(t1, t2));
        tf = fmin_(tf, fmax_(t1, t2));
    }
    /* the clip's own rounding: widen by the time it takes to travel 2 tol along the major axis */
    const float tslack = 2.0f * r.tol * fabsf(r.inv[r.w]);
    tn = fmax_(tn - tslack, 0.0f);
    tf = tf + tslack;
    if (!(tn <= tf)) return;
    r.tn = tn; r.tf = tf;
    const int64_t last2 = (int64_t)g->d2[r.w] - 1;
    if (r.sw > 0) walk_level(g, &r, 2, 0, last2); else walk_level(g, &r, 2, last2, 0);
    if (r.found) { *t_out = r.best; *idx_out = r.best_idx; }
    if (stats) { stats[0] += r.steps[0]; stats[1] += r.steps[1]; stats[2] += r.steps[2]; stats[3] += r.tests; }
}

/* ---- handle: mips built once per bitmask ------------------------------------------------------------------ */
walk_grid* vxo_walk_create(const uint32_t* words, const uint64_t dim[3], float vs, const float org[3])
{
    walk_grid* g = (walk_grid*)calloc(1, sizeof(walk_grid));
    g->words = words;
    g->vs = vs;
    g->half = vs * 0.5f;
    for (int a = 0; a < 3; ++a) {
        g->dim[a] = dim[a];
        g->org[a] = org[a];
        g->d1[a] = (dim[a] + 7) / 8;
        g->d2[a] = (g->d1[a] + 7) / 8;
    }
    const uint64_t n1 = g->d1[0] * g->d1[1] * g->d1[2], n2 = g->d2[0] * g->d2[1] * g->d2[2];
    g->m1 = (uint32_t*)calloc((size_t)(n1 / 32 + 2), 4);
    g->m2 = (uint32_t*)calloc((size_t)(n2 / 32 + 2), 4);
    const uint64_t nvox = dim[0] * dim[1] * dim[2], nwords = (nvox + 31) / 32;
    for (uint64_t wi = 0; wi < nwords; ++wi) {
        uint32_t wv = words[wi];
        while (wv) {
            const uint64_t i = wi * 32 + (uint64_t)__builtin_ctz(wv);
            wv &= wv - 1;
            const uint64_t x = i % dim[0], y = (i / dim[0]) % dim[1], z = i / (dim[0] * dim[1]);
            const uint64_t i1 = (x >> 3) + g->d1[0] * ((y >> 3) + g->d1[1] * (z >> 3));
            const uint64_t i2 = (x >> 6) + g->d2[0] * ((y >> 6) + g->d2[1] * (z >> 6));
            g->m1[i1 >> 5] |= 1u << (i1 & 31);
            g->m2[i2 >> 5] |= 1u << (i2 & 31);
        }
    }
    return g;
}

void vxo_walk_free(walk_grid* g)
{
    if (!g) return;
    free(g->m1);
    free(g->m2);
    free(g);
}

typedef struct { const walk_grid* g; const float* rays; uint64_t r0, r1; float tmin, tmax; float* t; uint64_t* idx; uint64_t stats[4]; } walk_arg;

static void* walk_main(void* p)
{
    walk_arg* a = (walk_arg*)p;
    for (uint64_t r = a->r0; r < a->r1; ++r) trace_one(a->g, a->rays + 6 * r, a->tmin, a->tmax, &a->t[r], &a->idx[r], a->stats);
    return NULL;
}

/* first hit per ray: t (-1 = miss) and the voxel index of the hit (~0 = miss); stats4 (optional): slabs looked at per level
 * (cells, bricks, blocks) and exact slab tests, summed over all rays */
void vxo_walk_trace(const walk_grid* g, const float* rays, uint64_t nrays, float tmin, float tmax, int threads, float* t_out, uint64_t* idx_out,
                    uint64_t* stats4)
{
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > nrays) threads = nrays ? (int)nrays : 1;
    walk_arg* wa = (walk_arg*)calloc((size_t)threads, sizeof(walk_arg));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    const uint64_t chunk = (nrays + (uint64_t)threads - 1) / (uint64_t)threads;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        const uint64_t b = (uint64_t)t * chunk;
        if (b >= nrays) break;
        walk_arg x;
        memset(&x, 0, sizeof(x));
        x.g = g; x.rays = rays; x.r0 = b; x.r1 = (b + chunk < nrays) ? b + chunk : nrays; x.tmin = tmin; x.tmax = tmax; x.t = t_out; x.idx = idx_out;
        wa[t] = x;
        pthread_create(&th[t], NULL, walk_main, &wa[t]);
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    if (stats4) {
        stats4[0] = stats4[1] = stats4[2] = stats4[3] = 0;
        for (int t = 0; t < started; ++t) for (int k = 0; k < 4; ++k) stats4[k] += wa[t].stats[k];
    }
    free(wa);
    free(th);
}
