"""GPU parity tests of the ray stage on the BASELINE configurations and on the code paths small grids never take:

* k_trace<false> (mips read from global memory: every grid above ~550^3) and the no-donation path (more than 2^32 voxels),
  forced on small grids through the library's environment switches AND exercised for real at 1024^3 / 2048^3;
* BASELINE configs[2] (the bench workload itself: atrium 512^3 + the bench's ray batch), configs[3] (atrium 1024^3) and
  configs[4] (10M-triangle soup, 2048^3, octree, 100M coherent primary rays) -- first-hit t and primitive id bit-equal to the
  oracle's brute force over ALL boxes on sampled rays, plus size-independent properties on every ray;
* rays with exactly-zero direction components (origins inside columns, on lattice / brick / block planes, and within an ulp
  of them).

The oracle (oracle/) is the checker; everything traced here goes through the C ABI of libvoxhip.so.
"""
import os

import numpy as np
import pytest

import oracle
import vx_scenes
from test_gpu_parity import axis_rays, check_trace, corner_rays, inside_rays

pytestmark = pytest.mark.gpu

NCORES = os.cpu_count() or 1


class env:
    """Set the library's run-time switches for the duration of a block (read at every launch)."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def aimed_rays(oa, gi, n, seed):
    """Rays from outside aimed at (jittered) centres of occupied boxes: a high hit rate on sparse grids."""
    rng = np.random.default_rng(seed)
    sel = rng.integers(0, len(oa), n)
    c = (oa["mn"][sel].astype(np.float64) + oa["mx"][sel].astype(np.float64)) * 0.5
    ext = (oa["mx"][0] - oa["mn"][0]).astype(np.float64)
    tgt = c + rng.uniform(-0.6, 0.6, (n, 3)) * ext
    lo, hi = gi["bmin"].astype(np.float64), gi["bmax"].astype(np.float64)
    R = 1.5 * np.linalg.norm(hi - lo)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = (lo + hi) / 2 + R * d
    dr = tgt - o
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    dr = dr.astype(np.float32)
    dr[dr == 0] = np.float32(1e-20)
    return np.ascontiguousarray(np.concatenate([o.astype(np.float32), dr], axis=1))


def zero_component_rays(oa, gi, vs, n, seed):
    """Axis-parallel rays (two zero components) and in-plane rays (one zero component), directions EXACTLY zero there.
    A third of the origins sits strictly inside a voxel column; a third exactly on a float plane of an occupied box (its own
    min or max, which is also a plane of the neighbouring box up to rounding), brick and block boundaries included; a third
    one ulp beside such a plane.  hitAabb (rint:46-56) yields NaN products there; the oracle and the kernel both resolve
    them with fminf/fmaxf, so the results must still be bit-equal."""
    rng = np.random.default_rng(seed)
    dim = np.array(gi["dim"])
    lo = gi["bmin"].astype(np.float64)
    hi = lo + dim * float(vs)
    rays = np.zeros((n, 6), np.float32)
    for i in range(n):
        b = oa[rng.integers(0, len(oa))]
        mode = i % 3
        nzero = 2 if (i // 3) % 2 == 0 else 1
        axes = rng.permutation(3)
        main = axes[0]
        p = np.zeros(3, np.float32)
        for a in range(3):
            if mode == 0:
                p[a] = np.float32(b["mn"][a] + (b["mx"][a] - b["mn"][a]) * rng.uniform(0.1, 0.9))
            else:
                plane = b["mn"][a] if rng.integers(0, 2) else b["mx"][a]
                if rng.integers(0, 4) == 0:   # a brick / block boundary plane, taken from the lattice the same way the boxes are
                    k = int(rng.integers(0, max(1, dim[a] // 8) + 1)) * (8 if rng.integers(0, 2) else 64)
                    k = min(k, int(dim[a]))
                    c = np.float32(gi["bmin"][a]) + (np.float32(k) + np.float32(0.5)) * np.float32(vs)
                    plane = np.float32(c - np.float32(vs) * np.float32(0.5))
                p[a] = np.float32(plane)
                if mode == 2:
                    p[a] = np.nextafter(p[a], np.float32(np.inf if rng.integers(0, 2) else -np.inf), dtype=np.float32)
        sgn = 1.0 if rng.integers(0, 2) else -1.0
        d = np.zeros(3, np.float32)
        d[main] = sgn
        if nzero == 1:
            d[axes[1]] = np.float32(rng.uniform(-1, 1))
        # origin outside the grid along the main axis (and, for in-plane rays, stepped back along the second axis accordingly)
        back = (p[main] - lo[main] + 3 * float(vs)) if sgn > 0 else (hi[main] - p[main] + 3 * float(vs))
        o = p.astype(np.float64) - d.astype(np.float64) * back / abs(float(d[main]))
        o32 = o.astype(np.float32)
        for a in range(3):
            if d[a] == 0:
                o32[a] = p[a]          # the constant coordinates keep their exact float
        rays[i, :3], rays[i, 3:] = o32, d
    assert (rays[:, 3:] == 0).sum() >= n
    return rays


# ---------------------------------------------------------------------------------------------- forced code paths, small grids
@pytest.mark.parametrize("lds,donate", [(0, 1), (1, 0), (0, 0)])
@pytest.mark.parametrize("name,vs", [("adversarial", 0.0625), ("blob70k", 2.0 / 64), ("soup2000", 0.02)])
def test_trace_forced_paths(gpu, name, vs, lds, donate):
    """k_trace<false> (VOXHIP_TRACE_LDS=0) and no work donation (VOXHIP_TRACE_DONATE=0) on grids the brute force handles."""
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, t), vs)
    ow, _, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    with env(VOXHIP_TRACE_LDS=lds, VOXHIP_TRACE_DONATE=donate):
        check_trace(gpu, g, oa, vx_scenes.random_rays(6000, gi["bmin"], gi["bmax"], seed=31))
        check_trace(gpu, g, oa, corner_rays(gi, float(vs), 6000, 32))
        check_trace(gpu, g, oa, axis_rays(gi, float(vs), 3000, 33))
        check_trace(gpu, g, oa, inside_rays(gi, float(vs), 6000, 34))
        tt, nh = g.trace(vx_scenes.random_rays(3000, gi["bmin"], gi["bmax"], seed=35), want_prim=False)
        ot, _ = oracle.trace_brute(oa, vx_scenes.random_rays(3000, gi["bmin"], gi["bmax"], seed=35))
        assert np.array_equal(tt, ot)


def test_brick_kernel_fused_and_separate_mip(gpu):
    """A grid whose brick rows are multiples of 64 (512^3): the brick kernel writes the level-1 mip itself and leaves empty bricks
    unwritten -- also on top of the stale bricks of an earlier, different build in the same handle; VOXHIP_FUSE_MIP1=0 stores every
    brick and derives the mip separately.  Same rays, same answers (the fused path itself is checked against the oracle by C3)."""
    v, t = vx_scenes.scene("soup2000")
    ext = float((v.max(0) - v.min(0)).max())
    vs = np.float32(ext / 512)
    mesh = gpu.Mesh.from_arrays(v, t)
    rays = np.concatenate([vx_scenes.random_rays(20000, v.min(0), v.max(0), seed=5), vx_scenes.random_rays(20000, v.min(0) - 1, v.max(0) + 1, seed=6)])
    with env(VOXHIP_FUSE_MIP1=0):
        g0 = gpu.Grid.voxelize(mesh, vs)
        t0, p0, _ = g0.trace(rays)
    # the fused build runs in a handle that first held ANOTHER mesh at the same resolution: bricks of that build that are empty
    # now keep their old words and must never be looked at
    v2 = (v + np.float32(0.013) * ext).astype(np.float32)
    v2 = np.minimum(np.maximum(v2, v.min(0)), v.max(0)).astype(np.float32)
    v2[0], v2[1] = v.min(0), v.max(0)          # same bounding box, hence the same grid
    g1 = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v2, t), vs)
    g1.trace(rays[:1000])
    g1.revoxelize(mesh, vs)
    assert g1.describe()["dim"] == g0.describe()["dim"] and np.array_equal(g1.bitmask(), g0.bitmask())
    t1, p1, _ = g1.trace(rays)
    assert np.array_equal(t0, t1) and np.array_equal(p0, p1) and (t1 > 0).sum() > 1000


# ---------------------------------------------------------------------------------------------- exact-zero direction components
@pytest.mark.parametrize("name,vs", [("cube", 0.0625), ("rotcube", 0.031), ("adversarial", 0.0625), ("blob70k", 2.0 / 128), ("soup2000", 1.0 / 256)])
def test_trace_zero_direction_components(gpu, name, vs):
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, t), vs)
    ow, _, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    rays = zero_component_rays(oa, gi, vs, 9000, 41)
    hit = check_trace(gpu, g, oa, rays)
    assert (hit[0::3] > 0).mean() > 0.5          # origins inside an occupied box's column: the axis-parallel half must hit it
    with env(VOXHIP_TRACE_LDS=0, VOXHIP_TRACE_DONATE=0):
        check_trace(gpu, g, oa, rays)
    # the shadow query walks the same cells
    sh = g.trace_ex(rays, any_hit=True, want=("shadowed",))["shadowed"]
    assert np.array_equal(sh, oracle.trace_any_brute(oa, rays))


# ---------------------------------------------------------------------------------------------- big sparse grids: the real thing
@pytest.mark.parametrize("res", [1024, 2048])
def test_trace_sparse_soup_large_grid(gpu, res):
    """soup2000 at 1024^3 (k_trace<false>: the level-1 mip no longer fits LDS) and at 2048^3 (additionally more than 2^32
    voxels: 64-bit voxel indices, no work donation) -- few occupied boxes, so the brute force over all of them is cheap."""
    v, t = vx_scenes.scene("soup2000")
    vs = np.float32(1.0 / res)
    mesh = gpu.Mesh.from_arrays(v, t)
    g = gpu.Grid.voxelize(mesh, vs)
    ow, calls, gi = oracle.build_bool(v, t, vs, threads=min(NCORES, 32))
    d = g.describe()
    assert d["dim"] == gi["dim"] and min(gi["dim"]) > res - 8
    if res == 2048:
        assert int(np.prod(gi["dim"])) > 2 ** 32
    assert np.array_equal(g.bitmask(), ow) and d["set_calls"] == calls
    oa = oracle.bool_aabbs(ow, gi, vs)
    assert d["occupied"] == len(oa)
    n = 5000
    hit = check_trace(gpu, g, oa, aimed_rays(oa, gi, n, 51))
    assert (hit > 0).mean() > 0.5
    check_trace(gpu, g, oa, vx_scenes.random_rays(n, gi["bmin"], gi["bmax"], seed=52))
    check_trace(gpu, g, oa, corner_rays(gi, float(vs), n, 53))
    check_trace(gpu, g, oa, axis_rays(gi, float(vs), 2000, 54))
    check_trace(gpu, g, oa, inside_rays(gi, float(vs), n, 55))
    check_trace(gpu, g, oa, zero_component_rays(oa, gi, vs, 3000, 56))
    out = g.trace_ex(aimed_rays(oa, gi, 2000, 57), want=("t", "prim", "normal"))
    ot, op = oracle.trace_brute(oa, aimed_rays(oa, gi, 2000, 57))
    assert np.array_equal(out["t"], ot) and np.array_equal(out["prim"], op)
    assert np.array_equal(out["normal"], oracle.cube_normals(oa, op, aimed_rays(oa, gi, 2000, 57), ot))


def formula_t(oa, prim, rays):
    """rint:46-56 evaluated in numpy float32 on the reported primitive's own box."""
    o, d = rays[:, :3], rays[:, 3:]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.float32(1.0) / d
        b = oa[prim]
        tb, tp = inv * (b["mn"] - o), inv * (b["mx"] - o)
        return np.fmax(np.fmax(np.fmin(tb, tp)[:, 0], np.fmin(tb, tp)[:, 1]), np.fmin(tb, tp)[:, 2]).astype(np.float32)


# ---------------------------------------------------------------------------------------------- BASELINE configs[2]: the bench workload
def test_c3_bench_workload_ray_parity(gpu):
    """atrium262k at exactly 512^3 with bench.py's own ray batch (1M random rays, seed 2): the whole batch is traced as the
    bench traces it and compared in full (t and primitive bit-equal) with the oracle's grid walk; 3000 sampled rays of it plus
    3000 rays starting INSIDE the hall are compared with the brute force over all occupied boxes (the definition), and every
    hit of the batch must be the rint formula of its reported box."""
    v, t = vx_scenes.scene("atrium262k")
    vs = np.float32((v.max(0) - v.min(0)).max() / 512)
    mesh = gpu.Mesh.from_arrays(v, t)
    g = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC)
    ow, calls, gi = oracle.build_bool(v, t, vs, threads=min(NCORES, 32))
    assert gi["dim"] == (512, 512, 512) and np.array_equal(g.bitmask(), ow)
    oa = oracle.bool_aabbs(ow, gi, vs)
    rays = vx_scenes.random_rays(1_000_000, v.min(0), v.max(0), seed=2)
    tt, pp, nh = g.trace(rays)
    sel = np.random.default_rng(61).choice(len(rays), 3000, replace=False)
    ot, op = oracle.trace_brute(oa, rays[sel], threads=NCORES)
    assert np.array_equal(tt[sel], ot) and np.array_equal(pp[sel], op)
    # 100 % of the batch against the oracle's grid-walking tracer (CPU-proven equal to the brute force; re-checked here on the sample)
    wt, wp = oracle.trace_walk(ow, gi, vs, rays, threads=NCORES)
    assert np.array_equal(wt[sel], ot) and np.array_equal(wp[sel], op)
    assert np.array_equal(tt, wt) and np.array_equal(pp, wp)
    h = np.flatnonzero(tt > 0)
    assert len(h) == nh and np.all(pp[tt <= 0] == 0xFFFFFFFF)
    assert np.array_equal(formula_t(oa, pp[h], rays[h]), tt[h])
    ins = inside_rays(gi, float(vs), 3000, 62)
    ti, pi, _ = g.trace(ins)
    oti, opi = oracle.trace_brute(oa, ins, threads=NCORES)
    assert np.array_equal(ti, oti) and np.array_equal(pi, opi)
    assert 0.2 < (oti > 0).mean()


def test_interior_camera_batch(gpu):
    """bench.py's `trace_interior` workload: the reference's camera model (raytrace.rgen:41-51) standing INSIDE the hall, 1280 x 720.
    Every pixel's ray, given explicitly, must trace bit-equal to the oracle's grid walk (and a sample to the brute force); the
    in-kernel generated rays must agree on hit / miss and on t within the north-star tolerance (1e-5)."""
    import torch
    v, t = vx_scenes.scene("atrium262k")
    vs = np.float32(32.0 / 512)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, t), vs)
    ow, _, gi = oracle.build_bool(v, t, vs, threads=min(NCORES, 32))
    assert np.array_equal(g.bitmask(), ow)
    oa = oracle.bool_aabbs(ow, gi, vs)
    W, H = 1280, 720
    dev = torch.device("cuda", 0)
    for cam in vx_scenes.INTERIOR_CAMERAS:
        vi, pi = vx_scenes.camera_matrices(**cam)
        rays = oracle.primary_rays(vi, pi, W, H)
        te, pe, _ = g.trace(rays)
        wt, wp = oracle.trace_walk(ow, gi, vs, rays, threads=NCORES)
        assert np.array_equal(te, wt) and np.array_equal(pe, wp)
        sel = np.random.default_rng(91).choice(W * H, 1500, replace=False)
        bt, bp = oracle.trace_brute(oa, rays[sel], threads=NCORES)
        assert np.array_equal(te[sel], bt) and np.array_equal(pe[sel], bp)
        d_t = torch.empty(W * H, dtype=torch.float32, device=dev)
        d_p = torch.empty(W * H, dtype=torch.int32, device=dev)
        g.trace_primary_device(vi, pi, W, H, d_t.data_ptr(), d_p.data_ptr())
        torch.cuda.synchronize()
        tk = d_t.cpu().numpy()
        assert np.array_equal(tk > 0, te > 0) and np.allclose(tk, te, rtol=0, atol=1e-5)
        assert (te > 0).mean() > 0.8            # the camera looks at the back of the hall: nearly every pixel sees a voxel


# ---------------------------------------------------------------------------------------------- BASELINE configs[3]: 1024^3
def test_c4_atrium_1024_rays(gpu):
    """atrium262k at exactly 1024^3 traces through k_trace<false>; sampled random / inside / corner rays against the brute
    force over all ~10M boxes (the occupancy itself is checked by test_c4_atrium_1024_eight_shards)."""
    v, t = vx_scenes.scene("atrium262k")
    vs = np.float32(32.0 / 1024)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, t), vs)
    ow, calls, gi = oracle.build_bool(v, t, vs, threads=min(NCORES, 32))
    assert gi["dim"] == (1024, 1024, 1024) and np.array_equal(g.bitmask(), ow)
    oa = oracle.bool_aabbs(ow, gi, vs)
    rays = np.concatenate([vx_scenes.random_rays(1500, gi["bmin"], gi["bmax"], seed=71), inside_rays(gi, float(vs), 1500, 72),
                           corner_rays(gi, float(vs), 1000, 73), zero_component_rays(oa, gi, vs, 600, 74)])
    tt, pp, _ = g.trace(rays)
    ot, op = oracle.trace_brute(oa, rays, threads=NCORES)
    assert np.array_equal(tt, ot) and np.array_equal(pp, op)
    # the throughput batch as well: every hit is the formula of its own box
    big = vx_scenes.random_rays(2_000_000, gi["bmin"], gi["bmax"], seed=75)
    tb, pb, nh = g.trace(big)
    h = np.flatnonzero(tb > 0)
    assert len(h) == nh and np.array_equal(formula_t(oa, pb[h], big[h]), tb[h])


# ---------------------------------------------------------------------------------------------- BASELINE configs[4]: 10M tris, 2048^3
def test_c5_soup_10m_2048_octree_and_primary_rays(gpu):
    """Synthetic 10M-triangle soup at 2048^3: VoxelGridBool occupancy, the Octree (sparse) path and 100M coherent primary rays
    (a 10000 x 10000 image from the reference camera model, raytrace.rgen:41-47, aimed so that most rays hit).  Occupancy and
    octree are compared IN FULL with the threaded oracle: all 2^28 bitmask words, the setVoxel call count, all ~75M sorted
    Morton items (the oracle's hit stream, Morton-coded and sorted with numpy -- sorting uint64 keys has one answer -- instead of
    the C restatement's qsort) and all ~15M 40-byte nodes (Octree::buildNodeRecursive restated, run on that sorted list).
    The rays are checked against the definition: brute force over all ~75M boxes on sampled rays (a skipped closer box would
    show), the oracle's grid walk on a 2M-ray sample, plus the formula property on a large sample."""
    import torch
    NT, G = 10_000_000, 2048
    v, t = vx_scenes.soup(NT, seed=4, edge=1.5 / G)
    vs = np.float32(1.0 / G)
    dev = torch.device("cuda", 0)
    dv, dt_ = torch.from_numpy(v).to(dev), torch.from_numpy(t).to(dev)
    mesh = gpu.Mesh.from_device(dv.data_ptr(), len(v), dt_.data_ptr(), len(t), keep=(dv, dt_))
    torch.cuda.synchronize()
    g = gpu.Grid.voxelize(mesh, vs, gpu.GRID_BOOL)
    d = g.describe()
    gi = oracle.grid_info(v, vs)
    assert d["dim"] == gi["dim"] and int(np.prod(d["dim"])) > 2 ** 32
    words = g.bitmask()
    assert int(np.bitwise_count(words).sum()) == d["occupied"]
    # ---- occupancy in full: the oracle's hit stream of ALL triangles (threaded driver, SAT a7 as the Octree uses it; the two SAT
    # variants give the same verdicts, I1) -> bitmask words and call count
    nthr = min(NCORES, 64)
    h = oracle.hits(v, t, vs, threads=nthr, sat=0)
    assert len(h) == d["set_calls"]
    X, Y = np.uint64(d["dim"][0]), np.uint64(d["dim"][1])
    idx = h[:, 0].astype(np.uint64) + X * (h[:, 1].astype(np.uint64) + Y * h[:, 2].astype(np.uint64))
    exp = np.zeros_like(words)
    # (bitwise_or.at is slow on 75M entries: sort the voxel indices, OR the bits of equal words with reduceat)
    idx.sort()
    wi = (idx >> np.uint64(5)).astype(np.int64)
    bits = np.uint32(1) << (idx & np.uint64(31)).astype(np.uint32)
    first = np.flatnonzero(np.r_[True, wi[1:] != wi[:-1]])
    exp[wi[first]] = np.bitwise_or.reduceat(bits, first)
    assert np.array_equal(words, exp), "occupancy bitmask differs from the oracle's (all 10M triangles)"
    del exp, wi, bits, first, idx
    # ---- octree in full: items and nodes byte for byte
    o = gpu.Octree(mesh, vs)
    assert o.num_items == d["set_calls"]
    items = o.items()
    oitems = oracle.morton3d_np(h[:, 0], h[:, 1], h[:, 2])
    del h
    oitems.sort()
    assert np.array_equal(items, oitems), "sorted Morton items differ from the oracle's"
    assert int((np.diff(items) != 0).sum()) + 1 == d["occupied"]
    assert o.memory_bytes() == 8 * o.num_items + 40 * o.num_nodes
    nodes = o.nodes()
    onodes = oracle.octree_nodes_from_sorted_items(oitems, int(np.ceil(np.log2(max(d["dim"])))), 16)
    assert len(nodes) == len(onodes) and nodes.tobytes() == onodes.tobytes(), "octree nodes differ from the oracle's"
    del items, nodes, oitems, onodes
    # rays: 100M primary rays in one launch
    vi, pi = vx_scenes.camera_matrices(eye=(1.55, 1.25, -0.85), ctr=(0.5, 0.5, 0.5), fov_deg=38.0, aspect=1.0)
    W = H = 10000
    d_t = torch.empty(W * H, dtype=torch.float32, device=dev)
    d_p = torch.empty(W * H, dtype=torch.int32, device=dev)
    g.trace_primary_device(vi, pi, W, H, d_t.data_ptr(), d_p.data_ptr())
    torch.cuda.synchronize()
    hit_rate = float((d_t > 0).float().mean().item())
    assert hit_rate >= 0.5, hit_rate
    oa = oracle.bool_aabbs(words, gi, vs)
    assert len(oa) == d["occupied"]
    rng = np.random.default_rng(81)
    nsamp = min(max(NCORES, 64), 256)
    pix = rng.choice(W * H, nsamp, replace=False)
    rays = oracle.primary_rays_pixels(vi, pi, W, H, pix)   # the oracle's camera restatement for the sampled pixels only
    tt = d_t.cpu().numpy()[pix]
    pp = d_p.cpu().numpy().view(np.uint32)[pix]
    # in-kernel ray generation differs from the explicit rays by rounding of the normalisation only; compare the brute force with
    # a trace of the SAME explicit rays (bit-equal), and the in-kernel result with both within the north-star tolerance
    te, pe, _ = g.trace(rays)
    ot, op = oracle.trace_brute(oa, rays, threads=NCORES)
    assert np.array_equal(te, ot) and np.array_equal(pe, op)
    assert np.array_equal(tt > 0, ot > 0) and np.allclose(tt, ot, rtol=0, atol=1e-5)
    assert (pp == op).mean() > 0.98
    # a 2M-pixel sample of the image as explicit rays: GPU trace == the oracle's grid walk, bit for bit (the walk == the brute force
    # on the sample above)
    pix2 = np.sort(rng.choice(W * H, 2_000_000, replace=False)).astype(np.uint64)
    rays2 = oracle.primary_rays_pixels(vi, pi, W, H, pix2)
    t2, p2, _ = g.trace(rays2)
    wt, wp = oracle.trace_walk(words, gi, vs, rays2, threads=NCORES)
    assert np.array_equal(t2, wt) and np.array_equal(p2, wp)
    wts, wps = oracle.trace_walk(words, gi, vs, rays, threads=NCORES)
    assert np.array_equal(wts, ot) and np.array_equal(wps, op)
    del rays2, t2, p2, wt, wp
    # formula property on 200 000 hits of the image
    ta = d_t.cpu().numpy()
    hh = np.flatnonzero(ta > 0)
    hh = hh[:: max(1, len(hh) // 200000)]
    pa = d_p.cpu().numpy().view(np.uint32)[hh]
    rr = oracle.primary_rays_pixels(vi, pi, W, H, hh.astype(np.uint64))
    f = formula_t(oa, pa, rr)
    assert np.allclose(f, ta[hh], rtol=0, atol=1e-5)
