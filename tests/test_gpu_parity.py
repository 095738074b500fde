"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: occupancy bitmask, AABB bytes, Vec / octree order bit-exact; first-hit t within 1e-5 of the brute-force minimum
(in practice bit-equal: the kernel evaluates the same float formula on the same float box).
"""
import numpy as np
import pytest

import oracle
import vx_scenes

pytestmark = pytest.mark.gpu

CUBE_SIZES = [0.5, 0.3, 0.25, 0.2, 0.1, 0.0625, 0.05]


def check_bool(vx, v, t, vs, sat=0, kind=None):
    vs = np.float32(vs)
    kind = vx.GRID_BOOL if kind is None else kind
    mesh = vx.Mesh.from_arrays(v, t)
    g = vx.Grid.voxelize(mesh, vs, kind, sat_variant=sat)
    ow, calls, gi = oracle.build_bool(v, t, vs, threads=0 if sat == 0 else 2)
    d = g.describe()
    assert d["dim"] == gi["dim"]
    assert np.array_equal(d["bbox_min"].view(np.uint32), gi["bmin"].view(np.uint32))   # sign of zero included
    assert np.array_equal(d["bbox_max"].view(np.uint32), gi["bmax"].view(np.uint32))
    assert np.array_equal(d["bbox_center"], gi["center"])
    w = g.bitmask()
    assert np.array_equal(w, ow), "bitmask mismatch: %d differing words" % int((w != ow).sum())
    oa = oracle.bool_aabbs(ow, gi, vs)
    a = g.aabbs()
    assert a.tobytes() == oa.tobytes()
    assert d["occupied"] == len(oa) and d["set_calls"] == calls and d["triangles"] == len(t)
    return g, mesh, gi, oa


@pytest.mark.parametrize("vs", CUBE_SIZES)
@pytest.mark.parametrize("sat", [0, 1])
def test_cube_all_sizes(gpu, vs, sat):
    """BASELINE configs[0]: cube.obj at voxelsize 0.1 (and the knife-edge neighbours)."""
    v, t = vx_scenes.cube()
    g, _, gi, oa = check_bool(gpu, v, t, vs, sat)
    assert g.memory_bytes() == 4 * ((gi["dim"][0] * gi["dim"][1] * gi["dim"][2] + 31) // 32)


@pytest.mark.parametrize("name,vs", [("rotcube", 0.09), ("rotcube", 0.031), ("adversarial", 0.125), ("adversarial", 0.1),
                                     ("adversarial", 0.0625), ("adversarial", 0.05), ("adversarial", 0.03125),
                                     ("soup2000", 0.02), ("soup20000", 1.0 / 256), ("blob70k", 2.0 / 64), ("blob70k", 2.0 / 256)])
@pytest.mark.parametrize("sat", [0, 1])
def test_scenes_bool(gpu, name, vs, sat):
    v, t = vx_scenes.scene(name)
    check_bool(gpu, v, t, vs, sat)


def test_atrium_512_bool_and_vec(gpu):
    """BASELINE configs[2]: ~262k-triangle architectural scene at 512^3, Bool occupancy and the VecEncoding list."""
    v, t = vx_scenes.scene("atrium262k")
    vs = np.float32(32.0 / 512)
    g, mesh, gi, oa = check_bool(gpu, v, t, vs)
    assert gi["dim"] == (512, 512, 512)
    gv = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC)
    ov = oracle.build_vec(v, t, vs)
    av = gv.aabbs()
    assert len(av) == len(ov) and av.tobytes() == ov.tobytes()
    assert gv.memory_bytes() == 24 * len(ov)
    assert np.array_equal(gv.bitmask(), g.bitmask())


@pytest.mark.parametrize("name,vs", [("cube", 0.25), ("cube", 0.0625), ("rotcube", 0.09), ("adversarial", 0.1), ("soup2000", 0.02),
                                     ("blob70k", 2.0 / 64)])
def test_vec_and_aabbstruct(gpu, name, vs):
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    mesh = gpu.Mesh.from_arrays(v, t)
    gv = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC)
    ov = oracle.build_vec(v, t, vs)
    assert gv.aabbs().tobytes() == ov.tobytes()                     # order and duplicates
    assert gv.describe()["set_calls"] == len(ov)
    ga = gpu.Grid.voxelize(mesh, vs, gpu.GRID_AABBSTRUCT)
    oa, by = oracle.build_aabbstruct(v, t, vs)
    assert ga.aabbs().tobytes() == oa.tobytes()
    assert ga.memory_bytes() == by


@pytest.mark.parametrize("name,vs,max_items", [("cube", 0.25, 16), ("cube", 0.0625, 16), ("rotcube", 0.05, 16), ("rotcube", 0.05, 4),
                                               ("adversarial", 0.0625, 16), ("soup2000", 0.02, 16), ("blob70k", 2.0 / 64, 16),
                                               ("blob70k", 2.0 / 64, 1)])
def test_octree(gpu, name, vs, max_items):
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    mesh = gpu.Mesh.from_arrays(v, t)
    o = gpu.Octree(mesh, vs, max_items)
    r = oracle.octree(v, t, vs, max_items=max_items, threads=2)
    assert o.num_items == len(r["items"]) and o.num_nodes == len(r["nodes"])
    assert np.array_equal(o.items(), r["items"])
    assert o.nodes().tobytes() == r["nodes"].tobytes()
    assert o.aabbs().tobytes() == r["aabbs"].tobytes()
    assert o.memory_bytes() == r["bytes"]
    mn, mx = o.root_bounds()
    assert np.array_equal(mn, r["root_min"]) and np.array_equal(mx, r["root_max"])


@pytest.mark.parametrize("ntri,grid,max_items", [(20000, 256, 16), (200000, 512, 16), (200000, 512, 3), (60000, 256, 64), (60000, 256, 65),
                                                 (60000, 256, 1000), (20000, 128, 0), (5000, 64, 2), (3000, 16, 16), (40000, 256, 10 ** 9)])
def test_octree_node_array_large(gpu, ntri, grid, max_items):
    """Octree node arrays of 10^5 .. 10^6 items (the direct node build's galloping searches cross many workgroups; max_items <= 64
    takes the direct form, larger values the level-by-level form): items and all 40-byte nodes against Octree::buildNodeRecursive
    restated (run on the sorted items)."""
    v, t = vx_scenes.soup(ntri, seed=4, edge=1.5 / grid)
    vs = np.float32(1.0 / grid)
    o = gpu.Octree(gpu.Mesh.from_arrays(v, t), vs, max_items)
    h = oracle.hits(v, t, vs, threads=4, sat=0)
    oitems = np.sort(oracle.morton3d_np(h[:, 0], h[:, 1], h[:, 2]))
    assert np.array_equal(o.items(), oitems)
    gi = oracle.grid_info(v, vs)
    ref = oracle.octree_nodes_from_sorted_items(oitems, int(np.ceil(np.log2(max(gi["dim"])))), max_items)
    nodes = o.nodes()
    assert len(nodes) == len(ref) and nodes.tobytes() == ref.tobytes()
    assert o.memory_bytes() == 8 * len(oitems) + 40 * len(ref)


@pytest.mark.parametrize("bits", [1, 5, 8, 9, 10, 18, 33, 48, 63, 64])
def test_item_sort_direct(gpu, bits):
    """vx_sort.hip (replaces std::sort(par_unseq), octTree.hpp:363) on its own: every pass count (1..8 digits of 8-9 bits), sizes
    around the 4096-key tile, heavy duplicates, already sorted and reversed input -- against numpy's sort."""
    rng = np.random.default_rng(bits)
    hi = (1 << bits) - 1
    for n in (0, 1, 2, 63, 64, 65, 4095, 4096, 4097, 12289, 100_003, 2_500_000):
        k = rng.integers(0, hi, n, dtype=np.uint64, endpoint=True)
        if n > 1000:
            k[rng.integers(0, n, n // 3)] = k[0]                # a third of the keys identical
            k[rng.integers(0, n, n // 10)] = np.uint64(hi)      # ... and many at the upper end
        assert np.array_equal(gpu.sort_u64(k, bits), np.sort(k)), (bits, n)
    k = np.arange(50_000, dtype=np.uint64) & np.uint64(hi)
    assert np.array_equal(gpu.sort_u64(k, bits), np.sort(k))
    assert np.array_equal(gpu.sort_u64(k[::-1], bits), np.sort(k))


def long_thin_mesh(ncells=100_000, ntri=4000, seed=9):
    """A mesh 100 000 x 8 x 8 voxels long (voxel size 1): small triangles all along x, a few long ones spanning > 65535 cells."""
    rng = np.random.default_rng(seed)
    c = np.stack([rng.uniform(0.0, ncells, ntri), rng.uniform(0.5, 7.5, ntri), rng.uniform(0.5, 7.5, ntri)], 1)
    tri = (c[:, None, :] + rng.uniform(-1.7, 1.7, (ntri, 3, 3))).astype(np.float32)
    long_ = np.array([[[10.3, 1.2, 1.1], [99000.7, 6.5, 2.2], [70000.1, 2.0, 6.9]],
                      [[65530.2, 0.7, 7.2], [65541.9, 7.1, 0.4], [65536.0, 4.0, 4.0]],
                      [[200.5, 7.7, 0.3], [80123.4, 0.2, 7.6], [80124.4, 0.9, 7.1]]], np.float32)
    tri = np.concatenate([tri, long_])
    tri[:, :, 0] = np.clip(tri[:, :, 0], 0.0, float(ncells))
    tri[:, :, 1:] = np.clip(tri[:, :, 1:], 0.0, 8.0)
    tri[0, 0] = (0.0, 0.0, 0.0)                      # pin the bounding box: exactly ncells x 8 x 8 cells of size 1
    tri[1, 0] = (float(ncells), 8.0, 8.0)
    v = np.ascontiguousarray(tri.reshape(-1, 3))
    t = np.arange(len(v), dtype=np.int32).reshape(-1, 3)
    return v, t


def test_axis_above_65535_cells(gpu):
    """A 100 000 x 8 x 8-cell grid (the reference's dims are size_t, VoxelBuilder.hpp:347-349; its Octree takes up to 2^21 cells
    per axis, octTree.hpp:577-588): Bool, AABBstruct-sized, Vec and Octree bit-equal to the oracle, incl. the Morton low-16-bit
    quirk (octTree.hpp:211-218) for x >= 65536; beyond 2^21 cells the Octree reports the reference's Morton-bits error."""
    v, t = long_thin_mesh()
    vs = np.float32(1.0)
    mesh = gpu.Mesh.from_arrays(v, t)
    for sat in (0, 1):
        g = gpu.Grid.voxelize(mesh, vs, gpu.GRID_BOOL, sat_variant=sat)
        ow, calls, gi = oracle.build_bool(v, t, vs, threads=0 if sat == 0 else 2)
        assert gi["dim"] == (100_000, 8, 8) and g.describe()["dim"] == gi["dim"]
        assert np.array_equal(g.bitmask(), ow) and g.describe()["set_calls"] == calls
        assert g.aabbs().tobytes() == oracle.bool_aabbs(ow, gi, vs).tobytes()
    gv = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC)
    ov = oracle.build_vec(v, t, vs)
    assert gv.aabbs().tobytes() == ov.tobytes() and (ov["mn"][:, 0] > 65536).any()
    o = gpu.Octree(mesh, vs)
    r = oracle.octree(v, t, vs, threads=2)
    assert np.array_equal(o.items(), r["items"]) and o.nodes().tobytes() == r["nodes"].tobytes()
    assert o.aabbs().tobytes() == r["aabbs"].tobytes() and o.memory_bytes() == r["bytes"]
    # per-voxel material ids on the wide grid (the unit kernels of that pass decode the same ranges)
    recs = np.zeros(5, dtype=gpu.MATERIAL)
    recs["diffuse"][:, 0] = np.arange(5) / 8.0
    ids = (np.arange(len(t)) % 5).astype(np.int32)
    mesh.set_materials(recs, ids)
    tv, nvalues, _ = _value_ids(recs, ids)
    gm = gpu.Grid.voxelize(mesh, vs, gpu.GRID_BOOL, materials=True)
    oids, order = oracle.material_ids(v, t, vs, tv, nvalues)
    assert np.array_equal(gm.bitmask(), ow) and np.array_equal(gm.materials()[1], oids)
    # rays on the wide grid (the ray kernel's 32-bit-coordinate variants): bit-equal to the brute force over all boxes
    oa = oracle.bool_aabbs(ow, gi, vs)
    rng = np.random.default_rng(17)
    n = 6000
    o = np.stack([rng.uniform(-2000.0, 102000.0, n), rng.uniform(-30.0, 40.0, n), rng.uniform(-30.0, 40.0, n)], 1)
    tgt = np.stack([rng.uniform(0.0, 100000.0, n), rng.uniform(0.0, 8.0, n), rng.uniform(0.0, 8.0, n)], 1)
    tgt[: n // 3, 0] = o[: n // 3, 0] + rng.uniform(-300.0, 300.0, n // 3)       # steep rays across the thin axes
    tgt[n // 3: n // 2, 0] = rng.uniform(65400.0, 65700.0, n // 2 - n // 3)      # around cell 65536
    rays = np.concatenate([o, tgt - o], 1).astype(np.float32)
    rays = np.concatenate([rays, np.array([[-5.0, 4.1, 3.9, 1.0, 0.0, 0.0], [100005.0, 2.2, 6.1, -1.0, 0.0, 0.0], [70000.5, -3.0, 4.5, 0.0, 1.0, 0.0]], np.float32)])
    for grid_ in (g, gm):
        tt, pp, _ = grid_.trace(rays, tmax=1.0e6)
        ot, op = oracle.trace_brute(oa, rays, tmax=1.0e6)
        assert np.array_equal(tt, ot) and np.array_equal(pp, op) and (ot > 0).mean() > 0.3
        wt, wp = oracle.trace_walk(ow, gi, vs, rays, tmax=1.0e6)
        assert np.array_equal(wt, ot) and np.array_equal(wp, op)
    # the reference's own limit: more than 2^21 cells on an axis
    with pytest.raises(gpu.VxError) as ei:
        gpu.Octree(mesh, np.float32(100_000 / (2 ** 21 + 5000.0)))
    assert ei.value.status == 5   # VX_ERR_MORTON_BITS
    with pytest.raises(gpu.VxError) as ei:
        gpu.Grid.voxelize(mesh, np.float32(100_000 / (2 ** 21 + 5000.0)))
    assert ei.value.status == 8   # VX_ERR_CAPACITY


def test_empty_inputs(gpu):
    flat_v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(flat_v, tri), 0.1)
    d = g.describe()
    assert d["dim"][2] == 0 and d["occupied"] == 0 and len(g.aabbs()) == 0 and g.bitmask().size == 0
    v, _ = vx_scenes.cube()
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, np.zeros((0, 3), np.int32)), 0.25)
    assert g.describe()["dim"] == (8, 8, 8) and g.describe()["occupied"] == 0
    o = gpu.Octree(gpu.Mesh.from_arrays(v, np.zeros((0, 3), np.int32)), 0.25)
    assert o.num_nodes == 0 and len(o.aabbs()) == 0
    o = gpu.Octree(gpu.Mesh.from_arrays(flat_v, tri), 0.1)
    assert o.num_items == 0 and len(o.aabbs()) == 0


def test_obj_file_path(gpu, tmp_path):
    """The `<obj_path> <voxelsize>` entry: parse + voxelize == arrays + voxelize."""
    v, t = vx_scenes.rotated_cube()
    p = tmp_path / "rot.obj"
    vx_scenes.write_obj(str(p), v, t)
    g = gpu.Grid.voxelize(gpu.Mesh.load_obj(str(p)), 0.07)
    ow, _, gi = oracle.build_bool(v, t, 0.07)
    assert np.array_equal(g.bitmask(), ow)


def test_grid_setvoxel_api(gpu):
    """VoxelGrid ctor / setVoxel / getCorrds / bounds errors (voxelgrid.hpp:52-100, voxelgridBool.cpp:54-68)."""
    org = (0.5, -1.25, 3.0)
    for kind in (gpu.GRID_BOOL, gpu.GRID_AABBSTRUCT, gpu.GRID_VEC):
        g = gpu.Grid.create(kind, 5, 7, 3, 0.3, org)
        pts = [(0, 0, 0), (4, 6, 2), (2, 3, 1), (2, 3, 1)]
        for p in pts:
            g.set_voxel(*p)
        assert g.test_voxel(2, 3, 1) and not g.test_voxel(1, 1, 1)
        gi = dict(dim=(5, 7, 3), bmin=np.array(org, np.float32))
        words = np.zeros(4, np.uint32)
        for x, y, z in pts:
            i = x + 5 * (y + 7 * z)
            words[i // 32] |= np.uint32(1 << (i % 32))
        exp = oracle.bool_aabbs(words, gi, np.float32(0.3))
        a = g.aabbs()
        if kind == gpu.GRID_VEC:
            assert len(a) == 4 and a[2].tobytes() == a[3].tobytes()     # duplicates kept, insertion order
            assert set(map(bytes, a.view(np.uint8).reshape(-1, 24))) == set(map(bytes, exp.view(np.uint8).reshape(-1, 24)))
        else:
            assert a.tobytes() == exp.tobytes()
        c = g.coords(2, 3, 1)
        assert np.array_equal(c, np.array(org, np.float32) + (np.array([2, 3, 1], np.float32) + np.float32(0.5)) * np.float32(0.3))
        with pytest.raises(gpu.VxError) as e:
            g.set_voxel(5, 0, 0)
        assert e.value.status == 4 and e.value.message == "Index out of bounds"
        with pytest.raises(gpu.VxError):
            g.coords(0, 7, 0)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_word_shards_or_to_full_mask(gpu, world):
    """Multi-GPU partition on one device: each logical rank writes only its word range; the shards are word-disjoint
    and their concatenation equals the single-GPU bitmask."""
    v, t = vx_scenes.scene("adversarial")
    vs = np.float32(0.03125)
    mesh = gpu.Mesh.from_arrays(v, t)
    full = gpu.Grid.voxelize(mesh, vs).bitmask()
    acc = np.zeros_like(full)
    for r in range(world):
        b, e, _ = gpu.shard_words(full.size, r, world)
        w = gpu.Grid.voxelize(mesh, vs, words=(b, e)).bitmask()
        assert not w[:b].any() and not w[e:].any()
        assert np.array_equal(w[b:e], full[b:e])
        acc |= w
    assert np.array_equal(acc, full)


def test_triangle_shards_concat_to_vec_list(gpu):
    v, t = vx_scenes.scene("soup2000")
    vs = np.float32(0.02)
    mesh = gpu.Mesh.from_arrays(v, t)
    full = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC).aabbs()
    parts = []
    for r in range(3):
        b, e = gpu.shard_range(len(t), r, 3)
        parts.append(gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC, tris=(b, e)).aabbs())
    assert np.concatenate(parts).tobytes() == full.tobytes()


# ---------------------------------------------------------------------------------------------- rays
def corner_rays(gi, vs, n, seed):
    """Rays aimed exactly at voxel lattice corners / edge midpoints: the grazing cases of the DDA."""
    rng = np.random.default_rng(seed)
    dim = np.array(gi["dim"])
    bmin = gi["bmin"].astype(np.float64)
    ctr = bmin + dim * vs / 2
    R = 2.5 * np.linalg.norm(dim * vs)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = (ctr + R * d).astype(np.float32)
    k = rng.integers(0, dim + 1, size=(n, 3)).astype(np.float64)
    k[: n // 2] += rng.choice([0.0, 0.5], size=(n // 2, 3))
    tgt = (bmin + k * vs).astype(np.float32)
    dr = tgt.astype(np.float64) - o.astype(np.float64)
    dr = (dr / np.linalg.norm(dr, axis=1, keepdims=True)).astype(np.float32)
    dr[dr == 0] = np.float32(1e-20)
    return np.ascontiguousarray(np.concatenate([o, dr], axis=1))


def axis_rays(gi, vs, n, seed):
    """Nearly axis-parallel rays running inside / along lattice planes."""
    rng = np.random.default_rng(seed)
    dim = np.array(gi["dim"])
    bmin = gi["bmin"].astype(np.float64)
    rays = np.zeros((n, 6), np.float32)
    for i in range(n):
        a = i % 3
        k = rng.integers(0, dim + 1).astype(np.float64)
        if i % 2:
            k += 0.5
        o = bmin + k * vs
        o[a] = bmin[a] - 3.0 * vs * (1 + rng.uniform())
        d = rng.uniform(-1, 1, 3) * (1e-7 if i % 4 < 2 else 1e-3)
        d[a] = 1.0
        if i % 8 >= 4:
            o[a] = bmin[a] + (dim[a] + 3.0) * vs
            d[a] = -1.0
        rays[i, :3], rays[i, 3:] = o, d
    rays[:, 3:][rays[:, 3:] == 0] = np.float32(1e-20)
    return rays


def inside_rays(gi, vs, n, seed):
    rng = np.random.default_rng(seed)
    dim = np.array(gi["dim"])
    bmin = gi["bmin"].astype(np.float64)
    o = bmin + rng.uniform(0, 1, (n, 3)) * dim * vs
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    d[d == 0] = np.float32(1e-20)
    return np.ascontiguousarray(np.concatenate([o.astype(np.float32), d], axis=1))


def check_trace(vx, g, oa, rays, tmin=0.001, tmax=10000.0):
    t, p, nh = g.trace(rays, tmin, tmax)
    ot, op = oracle.trace_brute(oa, rays, tmin, tmax)
    bad = np.flatnonzero((t > 0) != (ot > 0))
    assert bad.size == 0, "hit/miss mismatch on rays %s: gpu %s oracle %s" % (bad[:5], t[bad[:5]], ot[bad[:5]])
    far = np.flatnonzero(np.abs(t - ot) > 1e-5)
    assert far.size == 0, "t off by > 1e-5 on %d rays, e.g. %s: gpu %s oracle %s rays %s" % (far.size, far[:4], t[far[:4]], ot[far[:4]], rays[far[:2]])
    assert np.array_equal(t, ot), "t not bit-equal (max diff %g)" % np.abs(t - ot).max()
    assert np.array_equal(p, op)
    assert nh == int((ot > 0).sum())
    return t


@pytest.mark.parametrize("name,vs", [("cube", 0.25), ("cube", 0.0625), ("rotcube", 0.09), ("adversarial", 0.0625), ("adversarial", 0.1),
                                     ("soup2000", 0.02), ("blob70k", 2.0 / 64)])
def test_trace_vs_brute_force(gpu, name, vs):
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    mesh = gpu.Mesh.from_arrays(v, t)
    g = gpu.Grid.voxelize(mesh, vs)
    ow, _, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    n = 20000 if len(oa) < 20000 else 6000
    hit = check_trace(gpu, g, oa, vx_scenes.random_rays(n, gi["bmin"], gi["bmax"], seed=2))
    assert (hit > 0).mean() > 0.02
    check_trace(gpu, g, oa, corner_rays(gi, float(vs), n, 5))
    check_trace(gpu, g, oa, axis_rays(gi, float(vs), 4000, 6))
    check_trace(gpu, g, oa, inside_rays(gi, float(vs), n, 7))


def test_trace_blob_256(gpu):
    """BASELINE configs[1]: ~70k-triangle closed mesh, 256^3, 1M random rays vs the CPU first-hit t.  ALL rays are compared (t and
    primitive bit-equal) with the oracle's grid-walking tracer, which tests/test_oracle.py proves equal to the brute force; a
    3000-ray sample is also compared with the brute force over all ~290k boxes itself (the definition)."""
    v, t = vx_scenes.scene("blob70k")
    vs = np.float32(2.0 / 256)
    g, mesh, gi, oa = check_bool(gpu, v, t, vs)
    rays = vx_scenes.random_rays(1_000_000, gi["bmin"], gi["bmax"], seed=2)
    tt, pp, nh = g.trace(rays)
    sel = np.random.default_rng(11).choice(len(rays), 3000, replace=False)
    ot, op = oracle.trace_brute(oa, rays[sel])
    assert np.array_equal(tt[sel], ot) and np.array_equal(pp[sel], op)
    ow, _, _ = oracle.build_bool(v, t, vs)
    wt, wp = oracle.trace_walk(ow, gi, vs, rays)
    assert np.array_equal(wt[sel], ot) and np.array_equal(wp[sel], op)          # the checker against the definition
    assert np.array_equal(tt, wt) and np.array_equal(pp, wp)                    # 100 % of the batch
    # size-independent property on all 1M rays: the reported t is the rint formula of the reported primitive's own box,
    # and no ray reports a t outside the interval
    h = np.flatnonzero(tt > 0)
    assert len(h) == nh and np.all(pp[tt <= 0] == 0xFFFFFFFF)
    o, d = rays[h, :3], rays[h, 3:]
    inv = np.float32(1.0) / d
    b = oa[pp[h]]
    tb, tp = inv * (b["mn"] - o), inv * (b["mx"] - o)
    t0 = np.minimum(tb, tp).max(axis=1)
    assert np.array_equal(t0.astype(np.float32), tt[h])
    assert np.all(tt[h] >= np.float32(0.001)) and np.all(tt[h] <= np.float32(10000.0))


def test_trace_interval_and_compaction(gpu):
    import torch
    v, t = vx_scenes.rotated_cube()
    vs = np.float32(0.09)
    mesh = gpu.Mesh.from_arrays(v, t)
    g = gpu.Grid.voxelize(mesh, vs)
    ow, _, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    rays = vx_scenes.random_rays(5000, gi["bmin"], gi["bmax"], seed=3)
    for tmin, tmax in ((0.001, 10000.0), (9.5, 10000.0), (0.001, 9.3), (9.2, 9.6)):
        check_trace(gpu, g, oa, rays, tmin, tmax)
    # device entry point with ballot/prefix hit compaction
    dr = torch.from_numpy(rays).cuda()
    dt = torch.empty(len(rays), dtype=torch.float32, device="cuda")
    dp = torch.empty(len(rays), dtype=torch.int32, device="cuda")
    dh = torch.zeros(len(rays) * 3, dtype=torch.int32, device="cuda")
    dn = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    g.trace_device(dr.data_ptr(), len(rays), dt.data_ptr(), dp.data_ptr(), dh.data_ptr(), dn.data_ptr())
    torch.cuda.synchronize()
    ot, op = oracle.trace_brute(oa, rays)
    n = int(dn.item())
    assert n == int((ot > 0).sum())
    hits = dh.cpu().numpy().view(np.uint32).reshape(-1, 3)[:n]
    order = np.argsort(hits[:, 0])
    hr = hits[order]
    exp = np.flatnonzero(ot > 0)
    assert np.array_equal(hr[:, 0], exp) and np.array_equal(hr[:, 1], op[exp]) and np.array_equal(hr[:, 2].view(np.float32), ot[exp])


def test_primary_rays_match_generated_rays(gpu):
    """raytrace.rgen camera model generated in-kernel == explicit rays from the oracle's restatement of it."""
    import torch
    v, t = vx_scenes.rotated_cube(half=1.0, offset=(0.0, 1.0, 0.0))
    vs = np.float32(0.05)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, t), vs)
    vi, pi = vx_scenes.camera_matrices()
    W, H = 320, 180
    rays = oracle.primary_rays(vi, pi, W, H)
    t_explicit, _, _ = g.trace(rays)
    dt = torch.empty(W * H, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    g.trace_primary_device(vi, pi, W, H, dt.data_ptr())
    torch.cuda.synchronize()
    tp = dt.cpu().numpy()
    assert (t_explicit > 0).mean() > 0.01
    assert np.array_equal(tp > 0, t_explicit > 0)
    assert np.allclose(tp, t_explicit, rtol=0, atol=1e-5)


def test_revoxelize_reuses_handle(gpu):
    v, t = vx_scenes.rotated_cube()
    mesh = gpu.Mesh.from_arrays(v, t)
    g = gpu.Grid.voxelize(mesh, 0.09)
    a = g.bitmask().copy()
    g.revoxelize(mesh, 0.05)
    ow, _, gi = oracle.build_bool(v, t, 0.05)
    assert np.array_equal(g.bitmask(), ow)
    g.revoxelize(mesh, 0.09)
    assert np.array_equal(g.bitmask(), a)       # idempotent


def test_handle_reuse_sequence(gpu):
    """One handle through a sequence of rebuilds of different meshes / sizes / shards, queries in varying order.

    The handle keeps self-cleaning device state between builds (bbox reduction, scan status words), counts arrive through a
    pinned-host mailbox, and the traversal structures + word prefix are queued eagerly behind an unsharded build: every
    rebuild has to start from a clean state and every query has to see the build it follows."""
    scenes = [("rotcube", 0.09), ("soup2000", 0.02), ("cube", 0.2), ("blob70k", 2.0 / 64), ("cube", 0.0625), ("adversarial", 0.1), ("soup2000", 0.031)]
    g = None
    for k, (name, vs) in enumerate(scenes):
        v, t = vx_scenes.scene(name)
        vs = np.float32(vs)
        mesh = gpu.Mesh.from_arrays(v, t)
        if g is None:
            g = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC)
        elif k % 3 == 2:
            # two word shards into the same handle, OR-ed on the host (no eager structures: the mask is incomplete)
            nwords = (int(np.prod(oracle.grid_info(v, vs)["dim"])) + 31) // 32
            parts = []
            for r in range(2):
                wb, we, _ = gpu.shard_words(nwords, r, 2)
                g.revoxelize(mesh, vs, words=(wb, we))
                parts.append(g.bitmask().copy())
            ow, _, gi = oracle.build_bool(v, t, vs)
            assert np.array_equal(parts[0] | parts[1], ow)
            continue
        else:
            g.revoxelize(mesh, vs)
        ow, calls, gi = oracle.build_bool(v, t, vs)
        oa = oracle.bool_aabbs(ow, gi, vs)
        rays = vx_scenes.random_rays(3000, gi["bmin"], gi["bmax"], seed=10 + k)
        if k % 2 == 0:   # trace first (needs bricks + prefix), then the lists and the counts
            if len(oa):
                check_trace(gpu, g, oa, rays)
            assert np.array_equal(g.bitmask(), ow)
            d = g.describe()
        else:            # counts first
            d = g.describe()
            assert np.array_equal(g.bitmask(), ow)
            if len(oa):
                check_trace(gpu, g, oa, rays)
        assert d["occupied"] == len(oa) and d["set_calls"] == calls and d["dim"] == gi["dim"]
        ov = oracle.build_vec(v, t, vs)
        assert g.aabbs().tobytes() == ov.tobytes()     # Vec flavour: ordered list with duplicates
    # host-side setVoxel after a device build invalidates prefix / traversal structures
    v, t = vx_scenes.scene("cube")
    mesh = gpu.Mesh.from_arrays(v, t)
    gb = gpu.Grid.voxelize(mesh, np.float32(0.25))
    n0 = gb.describe()["occupied"]
    free = np.flatnonzero(np.unpackbits(gb.bitmask().view(np.uint8), bitorder="little")[: int(np.prod(gb.describe()["dim"]))] == 0)
    X, Y, Z = gb.describe()["dim"]
    i = int(free[0])
    gb.set_voxel(i % X, (i // X) % Y, i // (X * Y))
    assert gb.describe()["occupied"] == n0 + 1
    assert len(gb.aabbs()) == n0 + 1


# ---------------------------------------------------------------------------------------------- randomised parity
def test_random_soups_property(gpu):
    """Property test (seeded, the oracle is the checker): random triangle soups with random sizes, offsets (grid origins with
    both signs, non-multiple-of-32 dimensions) and voxel sizes -- bitmask, Vec order and octree items must match."""
    rng = np.random.default_rng(20260104)
    for case in range(24):
        n = int(rng.integers(1, 400))
        scale = float(10.0 ** rng.uniform(-2, 2))
        off = rng.uniform(-3, 3, 3) * scale
        v = (rng.uniform(0, 1, (3 * n, 3)) * scale * rng.uniform(0.05, 1.0, 3) + off).astype(np.float32)
        t = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
        if case % 3 == 0:   # shared vertices, fan
            t = np.stack([np.zeros(n, np.int32), rng.integers(0, 3 * n, n).astype(np.int32), rng.integers(0, 3 * n, n).astype(np.int32)], 1)
        ext = float((v.max(0) - v.min(0)).max())
        vs = np.float32(ext / float(rng.integers(3, 70)))
        mesh = gpu.Mesh.from_arrays(v, t)
        sat = case % 2
        g = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC, sat_variant=sat)
        ow, calls, gi = oracle.build_bool(v, t, vs, threads=0, sat=sat)
        assert g.describe()["dim"] == gi["dim"], (case, g.describe()["dim"], gi["dim"])
        assert np.array_equal(g.bitmask(), ow), "case %d bitmask" % case
        assert g.aabbs().tobytes() == oracle.build_vec(v, t, vs, threads=0, sat=sat).tobytes(), "case %d vec" % case
        if sat == 0:
            mi = int(rng.integers(1, 20))
            o = gpu.Octree(mesh, vs, mi)
            r = oracle.octree(v, t, vs, max_items=mi, threads=1)
            assert np.array_equal(o.items(), r["items"]) and o.nodes().tobytes() == r["nodes"].tobytes(), "case %d octree" % case
        gb = gpu.Grid.voxelize(mesh, vs, gpu.GRID_BOOL, sat_variant=sat)
        oa = oracle.bool_aabbs(ow, gi, vs)
        assert gb.aabbs().tobytes() == oa.tobytes(), "case %d aabbs" % case
        if len(oa):
            rays = vx_scenes.random_rays(1500, gi["bmin"], gi["bmin"] + np.array(gi["dim"], np.float32) * vs, seed=case)
            tt, pp, _ = gb.trace(rays)
            ot, op = oracle.trace_brute(oa, rays)
            assert np.array_equal(tt, ot) and np.array_equal(pp, op), "case %d rays" % case


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 1023, 4097])
def test_tiny_ray_batches(gpu, n):
    """Batch sizes around the wave / workgroup / chunk boundaries of the persistent ray kernel (static first chunk, guided
    chunk size, drain phase from the first round on), with and without the compacted hit list."""
    v, t = vx_scenes.scene("rotcube")
    vs = np.float32(0.09)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, t), vs)
    ow, _, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    rays = vx_scenes.random_rays(n, gi["bmin"], gi["bmax"], seed=1000 + n)
    check_trace(gpu, g, oa, rays)
    tt, nh = g.trace(rays, want_prim=False)   # no primitive ids: the split rays are merged by k_merge_flags instead of k_rank
    ot, _ = oracle.trace_brute(oa, rays)
    assert np.array_equal(tt, ot) and nh == int((ot > 0).sum())


def test_far_from_origin(gpu):
    """Meshes far from the origin (coordinates 1e2..2e3 x their own size): float32 spacing is then a visible fraction of a voxel.
    The voxelizer must still match bit for bit (same roundings as the reference), and the ray kernel's tolerance, which scales
    with max |coordinate|, must still enclose every box the brute-force formula can report."""
    rng = np.random.default_rng(77)
    for case in range(10):
        n = int(rng.integers(20, 300))
        scale = float(10.0 ** rng.uniform(-1, 1))
        off = rng.choice([-1.0, 1.0], 3) * rng.uniform(100.0, 2000.0, 3) * scale
        v = (rng.uniform(0, 1, (3 * n, 3)) * scale + off).astype(np.float32)
        t = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
        ext = float((v.max(0) - v.min(0)).max())
        vs = np.float32(ext / float(rng.integers(4, 40)))
        mesh = gpu.Mesh.from_arrays(v, t)
        g = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC, sat_variant=case % 2)
        ow, calls, gi = oracle.build_bool(v, t, vs, threads=0, sat=case % 2)
        assert g.describe()["dim"] == gi["dim"], (case, g.describe()["dim"], gi["dim"])
        assert np.array_equal(g.bitmask(), ow), "case %d bitmask" % case
        assert g.aabbs().tobytes() == oracle.build_vec(v, t, vs, threads=0, sat=case % 2).tobytes(), "case %d vec" % case
        oa = oracle.bool_aabbs(ow, gi, vs)
        if len(oa):
            hi = gi["bmin"] + np.array(gi["dim"], np.float32) * vs
            rays = vx_scenes.random_rays(2000, gi["bmin"], hi, seed=100 + case)
            tt, pp, _ = g.trace(rays)
            ot, op = oracle.trace_brute(oa, rays)
            assert np.array_equal(tt > 0, ot > 0), "case %d hit/miss" % case
            assert np.array_equal(tt, ot) and np.array_equal(pp, op), "case %d rays" % case


def test_soup_200k_at_512(gpu):
    """Mid-size soup at full 512^3 resolution against the (threaded) oracle: bitmask + octree items + node array."""
    v, t = vx_scenes.soup(200_000, seed=9, edge=0.006)
    vs = np.float32(1.0 / 512)
    mesh = gpu.Mesh.from_arrays(v, t)
    g = gpu.Grid.voxelize(mesh, vs, sat_variant=1)
    ow, calls, gi = oracle.build_bool(v, t, vs, threads=16)
    assert np.array_equal(g.bitmask(), ow) and g.describe()["set_calls"] == calls
    o = gpu.Octree(mesh, vs)
    r = oracle.octree(v, t, vs, threads=16)
    assert np.array_equal(o.items(), r["items"]) and o.nodes().tobytes() == r["nodes"].tobytes()


# ---------------------------------------------------------------------------------------------- raytrace2.rchit consumers
@pytest.mark.parametrize("name,vs", [("rotcube", 0.09), ("adversarial", 0.0625), ("blob70k", 2.0 / 64)])
def test_shadow_query_and_normals(gpu, name, vs):
    """Shadow rays (terminate on first hit, per-ray tMax = distance to the light; raytrace2.rchit:103-122) and the cube-face
    normal (rchit:60-73) against the oracle's restatements."""
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, t), vs)
    ow, _, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    n = 6000
    hi = gi["bmin"] + np.array(gi["dim"], np.float32) * vs
    rays = np.concatenate([vx_scenes.random_rays(n, gi["bmin"], hi, seed=21), corner_rays(gi, float(vs), n, 22), inside_rays(gi, float(vs), n, 23)])
    # closest hit + normal
    out = g.trace_ex(rays, want=("t", "prim", "normal"))
    ot, op = oracle.trace_brute(oa, rays)
    assert np.array_equal(out["t"], ot) and np.array_equal(out["prim"], op)
    on = oracle.cube_normals(oa, op, rays, ot)
    assert np.array_equal(out["normal"], on)
    hitrows = ot > 0
    assert np.all(np.abs(out["normal"][hitrows]).sum(1) == 1.0) and not out["normal"][~hitrows].any()
    # shadow query with a scalar and with a per-ray tMax
    sh = g.trace_ex(rays, any_hit=True, want=("shadowed",))["shadowed"]
    assert np.array_equal(sh, oracle.trace_any_brute(oa, rays))
    assert np.array_equal(sh.astype(bool), ot > 0)                     # same interval: shadowed <=> a closest hit exists
    rng = np.random.default_rng(24)
    tm = np.where(ot > 0, ot * rng.choice([0.5, 0.999, 1.0, 1.001, 2.0], size=len(ot)).astype(np.float32), np.float32(50.0)).astype(np.float32)
    sh2 = g.trace_ex(rays, any_hit=True, tmax_per_ray=tm, want=("shadowed",))["shadowed"]
    assert np.array_equal(sh2, oracle.trace_any_brute(oa, rays, tmax_per_ray=tm))
    # per-ray tMax also applies to the closest-hit query
    t3 = g.trace_ex(rays, tmax_per_ray=tm, want=("t",))["t"]
    exp = np.array([oracle.trace_brute(oa, rays[i:i + 1], tmax=float(tm[i]))[0][0] for i in range(0, len(rays), 37)], np.float32)
    assert np.array_equal(t3[::37], exp)


def test_primary_rays_with_normals(gpu):
    v, t = vx_scenes.rotated_cube(half=1.0, offset=(0.0, 1.0, 0.0))
    vs = np.float32(0.05)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, t), vs)
    ow, _, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    vi, pi = vx_scenes.camera_matrices()
    W, H = 160, 90
    out = g.trace_ex(camera=(vi, pi, W, H), want=("t", "prim", "normal"))
    rays = oracle.primary_rays(vi, pi, W, H)
    ref = g.trace_ex(rays, want=("t", "prim", "normal"))
    assert np.array_equal(out["t"] > 0, ref["t"] > 0) and np.allclose(out["t"], ref["t"], rtol=0, atol=1e-5)
    same = out["prim"] == ref["prim"]
    assert same.mean() > 0.999 and np.array_equal(out["normal"][same], ref["normal"][same])


def test_c4_atrium_1024_eight_shards(gpu):
    """BASELINE configs[3] on one device: the atrium at exactly 1024^3, voxelized as 8 word-aligned shards (what the 8 ranks
    of a node do) -- shards are word-disjoint, their union equals the unsharded mask, and that mask equals the oracle's."""
    v, t = vx_scenes.scene("atrium262k")
    vs = np.float32(32.0 / 1024)
    mesh = gpu.Mesh.from_arrays(v, t)
    full = gpu.Grid.voxelize(mesh, vs)
    d = full.describe()
    assert d["dim"] == (1024, 1024, 1024)
    fw = full.bitmask()
    ow, calls, gi = oracle.build_bool(v, t, vs, threads=16)
    assert np.array_equal(fw, ow) and d["set_calls"] == calls
    acc = np.zeros_like(fw)
    total_calls = 0
    g = None
    for r in range(8):
        b, e, chunk = gpu.shard_words(fw.size, r, 8)
        g = gpu.Grid.voxelize(mesh, vs, words=(b, e))
        w = g.bitmask()
        assert not (acc & w).any()                      # disjoint supports: all-gather == OR == sum
        acc |= w
        total_calls += g.describe()["set_calls"]
    assert np.array_equal(acc, fw)
    assert total_calls == calls                         # every setVoxel call lands in exactly one shard's words


# ---------------------------------------------------------------------------------------------- per-voxel materials
def _material_scene(name, nmat, seed):
    """Scene + `nmat` material records, two of them equal under MaterialObj::operator== (they differ only in ior / dissolve), and a
    per-triangle id pattern with runs, interleaving and some faces without a material (-1)."""
    v, t = vx_scenes.scene(name)
    rng = np.random.default_rng(seed)
    recs = np.zeros(nmat, dtype=np.dtype([("ambient", np.float32, 3), ("diffuse", np.float32, 3), ("specular", np.float32, 3), ("transmittance", np.float32, 3),
                                          ("emission", np.float32, 3), ("shininess", np.float32), ("ior", np.float32), ("dissolve", np.float32),
                                          ("illum", np.int32), ("texture_id", np.int32)]))
    for k in range(nmat):
        recs[k]["diffuse"] = rng.uniform(0, 1, 3)
        recs[k]["ambient"] = rng.uniform(0, 0.2, 3)
        recs[k]["specular"] = 0.5
        recs[k]["shininess"] = 10 + k
        recs[k]["ior"], recs[k]["dissolve"], recs[k]["illum"], recs[k]["texture_id"] = 1.0, 1.0, 2, -1
    if nmat >= 3:
        recs[2] = recs[0]
        recs[2]["ior"], recs[2]["dissolve"] = 1.5, 0.5          # == recs[0] for operator== (ior and dissolve are not compared)
    n = len(t)
    ids = (np.arange(n) // max(1, n // (3 * nmat))) % nmat
    ids[rng.integers(0, n, n // 7)] = rng.integers(0, nmat, n // 7)
    ids[rng.integers(0, n, n // 11)] = -1
    ids[: n // 50] = -1 if seed % 2 else nmat - 1                # which material is met FIRST decides the index order
    return v, t, recs, ids.astype(np.int32)


def _value_ids(recs, ids):
    """Value ids as the library forms them: 0 = MaterialObj{}, then the records de-duplicated by operator== in record order."""
    keyf = lambda r: (tuple(r["ambient"]), tuple(r["diffuse"]), tuple(r["specular"]), tuple(r["transmittance"]), tuple(r["emission"]), float(r["shininess"]), int(r["illum"]), -1)
    default = ((np.float32(0.1),) * 3, (np.float32(1), np.float32(1), np.float32(0)), (np.float32(1),) * 3, (np.float32(0),) * 3,
               (np.float32(0), np.float32(0), np.float32(0.1)), 0.0, 0, -1)
    keys, rec_value, values = [default], [], [None]
    for r in recs:
        k = keyf(r)
        if k not in keys:
            keys.append(k)
            values.append(r)
        rec_value.append(keys.index(k))
    tv = np.array([rec_value[i] if i >= 0 else 0 for i in ids], np.int32)
    return tv, len(keys), values


@pytest.mark.parametrize("name,vs,nmat,seed", [("cube", 0.25, 3, 1), ("rotcube", 0.09, 4, 2), ("adversarial", 0.0625, 5, 3), ("soup2000", 0.02, 6, 4), ("blob70k", 2.0 / 64, 4, 5)])
def test_material_ids(gpu, name, vs, nmat, seed):
    """VX_VOXELIZE_MATERIALS: getMatrials() / getMatIdx() as the reference's commented-out plumbing would fill them
    (VoxelBuilder.hpp:375-395, voxelgrid.hpp:102-114): last setVoxel call wins per voxel (Bool / AABBstruct), one id per call (Vec),
    materials indexed in the order they are first used, equal materials (operator==) share an index."""
    v, t, recs, ids = _material_scene(name, nmat, seed)
    vs = np.float32(vs)
    mesh = gpu.Mesh.from_arrays(v, t)
    mesh.set_materials(recs, ids)
    tv, nvalues, values = _value_ids(recs, ids)
    for kind in (gpu.GRID_BOOL, gpu.GRID_AABBSTRUCT, gpu.GRID_VEC):
        g = gpu.Grid.voxelize(mesh, vs, kind, materials=True)
        mats, mid = g.materials()
        per_call = kind == gpu.GRID_VEC
        calls = g.describe()["set_calls"]
        oids, order = oracle.material_ids(v, t, vs, tv, nvalues, per_call=per_call, ncalls=calls)
        assert np.array_equal(mid, oids), "kind %d: material ids differ" % kind
        assert len(mid) == (calls if per_call else g.describe()["occupied"]) == len(g.aabbs())
        assert len(mats) == len(order)
        for k, vid in enumerate(order):
            exp = values[vid]
            if exp is None:   # MaterialObj{}
                assert mats[k]["diffuse"].tolist() == [1, 1, 0] and mats[k]["ambient"].tolist() == [np.float32(0.1)] * 3 and mats[k]["shininess"] == 0 and mats[k]["illum"] == 0
            else:
                assert mats[k]["diffuse"].tolist() == exp["diffuse"].tolist() and mats[k]["shininess"] == exp["shininess"] and mats[k]["ior"] == exp["ior"]
        assert np.array_equal(g.bitmask(), gpu.Grid.voxelize(mesh, vs, gpu.GRID_BOOL).bitmask())   # occupancy is unaffected
    # without the flag: the reference as it runs today -- nothing is recorded
    mats, mid = gpu.Grid.voxelize(mesh, vs).materials()
    assert len(mats) == 0 and len(mid) == 0
    # a mesh without materials: every face carries MaterialObj{}
    plain = gpu.Mesh.from_arrays(v, t)
    g = gpu.Grid.voxelize(plain, vs, materials=True)
    mats, mid = g.materials()
    assert (len(mats) == 1 and not mid.any() and len(mid) == g.describe()["occupied"]) or g.describe()["occupied"] == 0


@pytest.mark.parametrize("name,vs,nmat,seed,nranks", [("adversarial", 0.0625, 5, 3, 3), ("soup2000", 0.02, 6, 4, 2), ("blob70k", 2.0 / 64, 4, 5, 8)])
def test_material_ids_sharded(gpu, name, vs, nmat, seed, nranks):
    """VX_VOXELIZE_MATERIALS on shards (SURVEY 8(e) + 8(f)2): word shards by rank (Bool) and triangle shards (Vec).  A shard knows the
    last triangle per voxel of its own slab, but the index of a material is the order of its first use over the WHOLE build
    (voxelgrid.hpp:102-114): the shards report their first uses, the element-wise minimum finishes every shard, and the shards' id
    arrays in shard order are the unsharded build's getMatIdx()."""
    v, t, recs, ids = _material_scene(name, nmat, seed)
    vs = np.float32(vs)
    mesh = gpu.Mesh.from_arrays(v, t)
    mesh.set_materials(recs, ids)
    tv, nvalues, values = _value_ids(recs, ids)
    whole = gpu.Grid.voxelize(mesh, vs, gpu.GRID_BOOL, materials=True)
    wm, wid = whole.materials()
    oids, order = oracle.material_ids(v, t, vs, tv, nvalues)
    assert np.array_equal(wid, oids)
    # ---- word shards by rank
    shards = [gpu.Grid.voxelize(mesh, vs, gpu.GRID_BOOL, materials=True, shard=(k, nranks)) for k in range(nranks)]
    assert all(len(s.materials()[1]) == 0 for s in shards)          # pending: nothing is reported before the finish
    fu = np.stack([s.material_first_use() for s in shards])
    fmin = np.where((fu >= 0).any(0), np.where(fu >= 0, fu, np.iinfo(np.int64).max).min(0), -1)
    assert np.array_equal(fmin, whole.material_first_use())
    for s in shards:
        s.finish_materials(fmin)
    assert np.array_equal(np.concatenate([s.materials()[1] for s in shards]), oids)
    for s in shards:
        assert s.materials()[0].tobytes() == wm.tobytes()
    acc = np.zeros_like(whole.bitmask())
    for s in shards:
        acc |= s.bitmask()
    assert np.array_equal(acc, whole.bitmask())
    # ---- triangle shards (Vec: one id per setVoxel call, call order)
    wv = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC, materials=True)
    ovids, vorder = oracle.material_ids(v, t, vs, tv, nvalues, per_call=True, ncalls=wv.describe()["set_calls"])
    assert np.array_equal(wv.materials()[1], ovids)
    T = len(t)
    cuts = [T * k // nranks for k in range(nranks + 1)]
    tsh = [gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC, materials=True, tris=(cuts[k], cuts[k + 1])) for k in range(nranks) if cuts[k + 1] > cuts[k]]
    fu = np.stack([s.material_first_use() for s in tsh])
    fmin = np.where((fu >= 0).any(0), np.where(fu >= 0, fu, np.iinfo(np.int64).max).min(0), -1)
    for s in tsh:
        s.finish_materials(fmin)
    assert np.array_equal(np.concatenate([s.materials()[1] for s in tsh]), ovids)
    assert tsh[0].materials()[0].tobytes() == wv.materials()[0].tobytes()


def test_multi_context_steady_state(gpu):
    """vx_multi: the mesh resident on every rank's device, persistent grids and worker threads; rebuilds at changing
    voxel sizes, with and without materials, to one destination or all -- each equal to the single-GPU build and the oracle."""
    v, t, recs, ids = _material_scene("blob70k", 4, 5)
    mesh = gpu.Mesh.from_arrays(v, t)
    mesh.set_materials(recs, ids)
    tv, nvalues, values = _value_ids(recs, ids)
    mc = gpu.Multi(mesh, [0] * 4)
    for rep, (vs, mats, allg, sat) in enumerate([(2.0 / 64, False, False, 0), (2.0 / 128, True, False, 1), (2.0 / 64, True, True, 0), (2.0 / 96, False, True, 0),
                                                  (2.0 / 128, True, False, 0)]):
        vs = np.float32(vs)
        gs = mc.voxelize(vs, sat_variant=sat, materials=mats, all_gather=allg)
        assert len(gs) == (4 if allg else 1)
        ow, calls, gi = oracle.build_bool(v, t, vs, threads=0 if sat == 0 else 2)
        oa = oracle.bool_aabbs(ow, gi, vs)
        for g in gs:
            d = g.describe()
            assert d["dim"] == gi["dim"] and d["set_calls"] == calls and d["occupied"] == len(oa)
            assert np.array_equal(g.bitmask(), ow) and g.aabbs().tobytes() == oa.tobytes()
            if mats:
                oids, order = oracle.material_ids(v, t, vs, tv, nvalues, sat=sat)
                gm, gid = g.materials()
                assert np.array_equal(gid, oids) and len(gm) == len(order)
            else:
                assert len(g.materials()[1]) == 0
        rays = vx_scenes.random_rays(2000, gi["bmin"], gi["bmax"], seed=20 + rep)
        tt, pp, _ = gs[-1].trace(rays)
        ot, op = oracle.trace_brute(oa, rays)
        assert np.array_equal(tt, ot) and np.array_equal(pp, op)
    mc.free()
    # error paths: an unknown device, a flavour whose order needs triangle shards, a borrowed (device-only) mesh
    with pytest.raises(gpu.VxError):
        gpu.Multi(mesh, [0, 99])
    with pytest.raises(gpu.VxError) as ei:
        gpu.Multi(mesh, [0, 0], kind=gpu.GRID_VEC)
    assert ei.value.status == 9
    import torch
    dv, dt_ = torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda()
    with pytest.raises(gpu.VxError):
        gpu.Multi(gpu.Mesh.from_device(dv.data_ptr(), len(v), dt_.data_ptr(), len(t), keep=(dv, dt_)), [0, 0])


# ---------------------------------------------------------------------------------------------- multi-GPU entry of the C ABI
@pytest.mark.parametrize("nranks,all_gather", [(2, False), (3, True), (8, False)])
@pytest.mark.parametrize("name,vs", [("adversarial", 0.03125), ("blob70k", 2.0 / 128)])
def test_voxelize_multi_logical_ranks(gpu, name, vs, nranks, all_gather):
    """vx_voxelize_multi with logical ranks on the one device of the test box: word shards + peer copies give the single-GPU
    grid -- bitmask, counts, AABB list, and a grid that traces (prefix + traversal structure rebuilt from the gathered mask)."""
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    mesh = gpu.Mesh.from_arrays(v, t)
    ow, calls, gi = oracle.build_bool(v, t, vs)     # serial driver = SAT a7, as sat_variant 0 (the threaded driver's a8 has no epsilon skips)
    oa = oracle.bool_aabbs(ow, gi, vs)
    grids = gpu.Grid.voxelize_multi(mesh, vs, [0] * nranks, all_gather=all_gather)
    assert len(grids) == (nranks if all_gather else 1)
    rays = vx_scenes.random_rays(3000, gi["bmin"], gi["bmax"], seed=91)
    for g in grids:
        d = g.describe()
        assert d["dim"] == gi["dim"] and d["occupied"] == len(oa) and d["set_calls"] == calls and d["triangles"] == len(t)
        assert np.array_equal(g.bitmask(), ow)
        assert g.aabbs().tobytes() == oa.tobytes()
        check_trace(gpu, g, oa, rays)
    with pytest.raises(gpu.VxError):
        gpu.Grid.voxelize_multi(mesh, vs, [0, 0], kind=gpu.GRID_VEC)
    # more ranks than words, and a flat mesh (no words at all)
    v2, t2 = vx_scenes.cube()
    g2 = gpu.Grid.voxelize_multi(gpu.Mesh.from_arrays(v2, t2), np.float32(0.5), [0] * 5)[0]
    w2, _, _ = oracle.build_bool(v2, t2, np.float32(0.5))
    assert np.array_equal(g2.bitmask(), w2)
    flat = gpu.Grid.voxelize_multi(gpu.Mesh.from_arrays(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.int32)), 0.1, [0, 0])[0]
    assert flat.describe()["occupied"] == 0 and flat.describe()["dim"][2] == 0


def test_vec_list_bound_to_caller_buffer(gpu):
    """vx_grid_bind_aabbs_device: VoxelGridVec builds emit straight into the caller's device buffer (no copy in getAabbs); a buffer
    that is too small falls back to the grid's own storage; setVoxel appends and re-binding keep the list intact."""
    import torch
    v, t = vx_scenes.scene("soup2000")
    vs = np.float32(0.02)
    mesh = gpu.Mesh.from_arrays(v, t)
    ov = oracle.build_vec(v, t, vs)
    g = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC)
    buf = torch.zeros((len(ov) + 8) * 6, dtype=torch.float32, device="cuda")
    g.bind_aabbs_device(buf.data_ptr(), len(ov) + 8)
    g.revoxelize(mesh, vs)
    torch.cuda.synchronize()
    assert buf.cpu().numpy()[: len(ov) * 6].tobytes() == ov.tobytes()            # the build itself filled the caller's buffer
    assert g.aabbs_device(buf.data_ptr(), len(ov) + 8) == len(ov)
    assert g.aabbs().tobytes() == ov.tobytes()
    # appending by setVoxel migrates the list into the grid's own storage
    g.set_voxel(0, 0, 0)
    a = g.aabbs()
    assert len(a) == len(ov) + 1 and a[: len(ov)].tobytes() == ov.tobytes()
    # a binding that is too small: the build falls back, the list is still right
    small = torch.zeros(60, dtype=torch.float32, device="cuda")
    g.bind_aabbs_device(small.data_ptr(), 10)
    g.revoxelize(mesh, vs)
    assert g.aabbs().tobytes() == ov.tobytes()
    g.bind_aabbs_device(None, 0)
    g.revoxelize(mesh, np.float32(0.031))
    assert g.aabbs().tobytes() == oracle.build_vec(v, t, np.float32(0.031)).tobytes()


def test_vec_list_async(gpu):
    """VX_VOXELIZE_LIST_ASYNC: the build returns with the list's length; the records are written beside the next ray batch (side stream of
    the handle) or on the grid's stream by whoever reads the list first.  Every path ends with the oracle's list, byte for byte."""
    import torch
    v, t = vx_scenes.scene("blob70k")
    vs = np.float32(2.0 / 96)
    mesh = gpu.Mesh.from_arrays(v, t)
    ov = oracle.build_vec(v, t, vs)
    ow, _, gi = oracle.build_bool(v, t, vs)
    rays = vx_scenes.random_rays(20000, gi["bmin"], gi["bmax"], seed=5)
    d_rays = torch.from_numpy(rays).cuda()
    d_t = torch.zeros(len(rays), dtype=torch.float32, device="cuda")
    d_p = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
    g = gpu.Grid.voxelize(mesh, vs, gpu.GRID_VEC)
    t_ref, p_ref, _ = g.trace(rays)
    cap = len(ov) + 8
    buf = torch.zeros(cap * 6, dtype=torch.float32, device="cuda")
    g.bind_aabbs_device(buf.data_ptr(), cap)
    for rep in range(3):  # build -> getAabbs (count only) -> rays with the emission beside them -> the next build waits for it
        buf.zero_()
        torch.cuda.synchronize()
        g.revoxelize(mesh, vs, list_async=True)
        assert g.aabbs_device(buf.data_ptr(), cap) == len(ov)
        g.trace_device(d_rays.data_ptr(), len(rays), d_t.data_ptr(), d_p.data_ptr())
        g.list_wait()
        got = buf.cpu().numpy()  # (torch's copy is ordered behind the legacy stream = the grid's stream here)
        assert got[: len(ov) * 6].tobytes() == ov.tobytes(), rep
        assert np.array_equal(d_t.cpu().numpy(), t_ref) and np.array_equal(d_p.cpu().numpy().view(np.uint32), p_ref)
    # no ray batch in between: the first reader queues the emission
    buf.zero_()
    g.revoxelize(mesh, vs, list_async=True)
    assert g.aabbs().tobytes() == ov.tobytes()
    g.revoxelize(mesh, vs, list_async=True)
    g.list_wait()
    torch.cuda.synchronize()
    assert buf.cpu().numpy()[: len(ov) * 6].tobytes() == ov.tobytes()
    # an unqueued emission is dropped by the next build; setVoxel and re-binding resolve a pending one
    g.revoxelize(mesh, vs, list_async=True)
    g.revoxelize(mesh, vs, list_async=True)
    g.trace_device(d_rays.data_ptr(), len(rays), d_t.data_ptr(), d_p.data_ptr())
    g.set_voxel(0, 0, 0)
    a = g.aabbs()
    assert len(a) == len(ov) + 1 and a[: len(ov)].tobytes() == ov.tobytes()
    g.bind_aabbs_device(None, 0)
    g.revoxelize(mesh, vs, list_async=True)  # without a binding: the grid's own storage
    g.trace_device(d_rays.data_ptr(), len(rays), d_t.data_ptr(), d_p.data_ptr())
    assert g.aabbs().tobytes() == ov.tobytes()
    g.revoxelize(mesh, vs, list_async=True)
    g.trace_device(d_rays.data_ptr(), len(rays), d_t.data_ptr(), d_p.data_ptr())
    g.free()  # with the emission possibly still running
    torch.cuda.synchronize()


def test_bool_list_async(gpu):
    """vx_grid_aabbs_device_async: VoxelGridBool::getAabbs with the emission left to the next ray batch (side stream) or to the first call
    that reads the list or changes what it is made from; the caller's buffer always ends up with the oracle's list."""
    import torch
    v, t = vx_scenes.scene("blob70k")
    vs = np.float32(2.0 / 96)
    mesh = gpu.Mesh.from_arrays(v, t)
    ow, _, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    rays = vx_scenes.random_rays(20000, gi["bmin"], gi["bmax"], seed=6)
    d_rays = torch.from_numpy(rays).cuda()
    d_t = torch.zeros(len(rays), dtype=torch.float32, device="cuda")
    d_p = torch.zeros(len(rays), dtype=torch.int32, device="cuda")
    for kind in (gpu.GRID_BOOL, gpu.GRID_AABBSTRUCT):
        g = gpu.Grid.voxelize(mesh, vs, kind)
        t_ref, p_ref, _ = g.trace(rays)
        cap = len(oa) + 8
        buf = torch.zeros(cap * 6, dtype=torch.float32, device="cuda")
        for rep in range(2):  # build -> getAabbs (count now) -> rays with the emission beside them
            buf.zero_()
            torch.cuda.synchronize()
            g.revoxelize(mesh, vs)
            assert g.aabbs_device_async(buf.data_ptr(), cap) == len(oa)
            g.trace_device(d_rays.data_ptr(), len(rays), d_t.data_ptr(), d_p.data_ptr())
            g.list_wait()
            assert buf.cpu().numpy()[: len(oa) * 6].tobytes() == oa.tobytes(), (kind, rep)
            assert np.array_equal(d_t.cpu().numpy(), t_ref) and np.array_equal(d_p.cpu().numpy().view(np.uint32), p_ref)
        # no ray batch: list_wait, the next build, setVoxel each queue the emission first (with the mask and prefix it was asked for)
        buf.zero_()
        assert g.aabbs_device_async(buf.data_ptr(), cap) == len(oa)
        g.list_wait()
        assert buf.cpu().numpy()[: len(oa) * 6].tobytes() == oa.tobytes()
        buf.zero_()
        assert g.aabbs_device_async(buf.data_ptr(), cap) == len(oa)
        g.revoxelize(mesh, np.float32(2.0 / 64))  # another grid: the list asked for before must still be the old one
        torch.cuda.synchronize()
        assert buf.cpu().numpy()[: len(oa) * 6].tobytes() == oa.tobytes()
        g.revoxelize(mesh, vs)
        buf.zero_()
        assert g.aabbs_device_async(buf.data_ptr(), cap) == len(oa)
        g.set_voxel(0, 0, 0)
        torch.cuda.synchronize()
        assert buf.cpu().numpy()[: len(oa) * 6].tobytes() == oa.tobytes()
        g.free()


def test_rebuild_after_external_write_of_the_mask(gpu):
    """The multi-rank exchange writes the grid's bitmask from outside the library (vx_grid_bitmask_device_mut: RCCL all-gather, peer
    copies).  A rebuild in the same handle must not see any of it: the mask is cleared by the build itself (k_tri_setup's threads
    or a memset) before k_voxelize's plain loads decide which atomicOr requests to skip -- an external all-ones mask would
    otherwise make every request look redundant."""
    import torch
    v, t = vx_scenes.scene("blob70k")
    vs = np.float32(2.0 / 128)
    mesh = gpu.Mesh.from_arrays(v, t)
    ow, calls, gi = oracle.build_bool(v, t, vs)
    for kind in (gpu.GRID_BOOL, gpu.GRID_VEC):
        g = gpu.Grid.voxelize(mesh, vs, kind)
        nwords = g.describe()["num_words"]

        class View:
            __cuda_array_interface__ = {"shape": (nwords,), "typestr": "<i4", "data": (g.bitmask_device_ptr(mutable=True), False), "version": 3, "strides": None}
        for fill in (-1, 0x55555555, 0):
            torch.as_tensor(View(), device="cuda").fill_(fill)      # the "exchange": every word overwritten from outside
            torch.cuda.synchronize()
            g.revoxelize(mesh, vs)
            assert np.array_equal(g.bitmask(), ow), "fill %x, kind %d" % (fill & 0xFFFFFFFF, kind)
            for sh in ((0, 2), (1, 2)):   # and word shards of it, as a rank of a multi-GPU build rebuilds them
                torch.as_tensor(View(), device="cuda").fill_(fill)
                torch.cuda.synchronize()
                if kind == gpu.GRID_BOOL:
                    g.revoxelize(mesh, vs, shard=sh)
                    wb, we, _ = gpu.shard_words(nwords, sh[0], sh[1])
                    exp = np.zeros_like(ow)
                    exp[wb:we] = ow[wb:we]
                    assert np.array_equal(g.bitmask(), exp)
        g.revoxelize(mesh, vs)
        assert np.array_equal(g.bitmask(), ow) and g.describe()["set_calls"] == calls


def test_rebuilds_alternating_meshes_same_grid(gpu):
    """Forty rebuilds in ONE handle, alternating between a mesh and a subset of it that spans the same bounding box (same grid,
    same bitmask addresses, fewer bits), both flavours.  The voxelizer skips an atomicOr when a plain load already shows the bits
    set: a line of the PREVIOUS build surviving in a cache would show bits that the fresh mask does not have and drop them.  Also
    exercises the polled mailbox totals (every build has its own sequence tag) and the block table queued ahead of the unit total
    (the subset has fewer units than the table of the build before it)."""
    v, t = vx_scenes.scene("blob70k")
    vs = np.float32(2.0 / 128)
    keep = np.zeros(len(t), bool)
    keep[::3] = True
    # keep every triangle that touches an extreme vertex, so that both meshes have the same bounding box
    ext = set(np.argmin(v, axis=0).tolist() + np.argmax(v, axis=0).tolist())
    keep |= np.isin(t, list(ext)).any(axis=1)
    tb = t[keep]
    for kind in (gpu.GRID_BOOL, gpu.GRID_VEC):
        ref = []
        for tri in (t, tb):
            ow, calls, gi = oracle.build_bool(v, tri, vs)
            ref.append((ow, calls, gi))
        assert ref[0][2]["dim"] == ref[1][2]["dim"] and not np.array_equal(ref[0][0], ref[1][0])
        meshes = [gpu.Mesh.from_arrays(v, t), gpu.Mesh.from_arrays(v, tb)]
        g = gpu.Grid.voxelize(meshes[0], vs, kind)
        for k in range(40):
            which = (k + 1) % 2 if k % 5 else k % 2   # A B A B ... with an occasional repeat
            g.revoxelize(meshes[which], vs)
            ow, calls, gi = ref[which]
            assert np.array_equal(g.bitmask(), ow), "rebuild %d (mesh %d, kind %d)" % (k, which, kind)
            if k % 7 == 0:
                d = g.describe()
                assert d["set_calls"] == calls and d["occupied"] == int(np.unpackbits(ow.view(np.uint8)).sum())
