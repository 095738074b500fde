"""CPU tests of the oracle (oracle/vx_oracle.c): the survey-recorded reference outputs, the invariants SURVEY.md
Appendix A verified on the unmodified reference, and the committed self-generated regression vectors.

Formal status: PARITY UNPINNED beyond the survey anchors -- the reference ships no fixtures and cannot be compiled in
this image without stand-in headers (glm, tinyobjloader, <print>)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
import vx_scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def morton_np(xyz):
    return np.array([oracle.morton3d(*p) for p in xyz], dtype=np.uint64)


def test_survey_anchor_counts():
    """cube +-1 at seven voxel sizes, counts recorded from the unmodified reference."""
    with open(os.path.join(GOLD, "survey_anchors.json")) as fh:
        A = json.load(fh)["cube_pm1"]
    v, t = vx_scenes.cube()
    for i, vs in enumerate(A["voxel_sizes"]):
        w, calls, gi = oracle.build_bool(v, t, vs)
        assert len(oracle.bool_aabbs(w, gi, vs)) == A["bool_occupied"][i]
        assert len(oracle.build_vec(v, t, vs)) == A["vec_items"][i] == calls
        oc = oracle.octree(v, t, vs)
        assert len(oc["items"]) == A["octree_items"][i]
        if vs == 0.25:
            assert oc["bytes"] == A["octree_bytes_at_0.25"] and len(oc["nodes"]) == A["octree_nodes_at_0.25"]


def test_knife_edge_cube():
    """I7: +-1 cube -> 0 voxels at vs 0.1 / 0.2 (20^3, 10^3); only the three min-side faces at 0.25."""
    v, t = vx_scenes.cube()
    for vs, dim in ((0.1, 20), (0.2, 10)):
        w, _, gi = oracle.build_bool(v, t, vs)
        assert gi["dim"] == (dim, dim, dim) and not w.any()
    w, _, gi = oracle.build_bool(v, t, 0.25)
    a = oracle.bool_aabbs(w, gi, 0.25)
    assert len(a) == 169
    # every occupied voxel touches a min-side face: x==0 or y==0 or z==0
    idx = np.flatnonzero(np.unpackbits(w.view(np.uint8), bitorder="little")[:512])
    x, y, z = idx % 8, (idx // 8) % 8, idx // 64
    assert np.all((x == 0) | (y == 0) | (z == 0))


@pytest.mark.parametrize("name,vs", [("cube", 0.0625), ("rotcube", 0.09), ("adversarial", 0.125), ("adversarial", 0.1),
                                     ("adversarial", 0.05), ("soup2000", 0.02), ("soup2000", 0.013)])
def test_invariants_I1_to_I5(name, vs):
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    w, calls, gi = oracle.build_bool(v, t, vs)
    a = oracle.bool_aabbs(w, gi, vs)
    # I1: serial driver (SAT a7) == threaded driver (SAT a8), byte for byte
    # (SAT a7 skips axes/normals with L1 norm < 1e-8, a8 does not: on triangles with edges <= 1e-9 a7 may report MORE
    # hits, never fewer; the occupancy of these inputs is nevertheless identical, as the survey found on the reference)
    w2, calls2, _ = oracle.build_bool(v, t, vs, threads=3)
    assert np.array_equal(w, w2)
    assert calls == calls2 if name != "adversarial" else calls >= calls2
    # driver and SAT are independent choices: threaded driver + a7 == serial driver + a7, call for call
    w3, calls3, _ = oracle.build_bool(v, t, vs, threads=3, sat=0)
    assert np.array_equal(w, w3) and calls3 == calls
    w4, calls4, _ = oracle.build_bool(v, t, vs, threads=0, sat=1)
    assert np.array_equal(w2, w4) and calls4 == calls2
    # I2: AABBstruct list identical to Bool list; memory = 28 * N
    a2, by = oracle.build_aabbstruct(v, t, vs)
    assert a2.tobytes() == a.tobytes()
    assert by == 28 * gi["dim"][0] * gi["dim"][1] * gi["dim"][2]
    # I3: unique(Vec) == Bool as sets
    vec = oracle.build_vec(v, t, vs)
    assert len(vec) == calls
    assert set(map(bytes, np.unique(vec).view(np.uint8).reshape(-1, 24))) == set(map(bytes, a.view(np.uint8).reshape(-1, 24)))
    # I4: Octree::getAabbs == Vec stably sorted by Morton(x,y,z), duplicates included
    h = oracle.hits(v, t, vs)
    assert len(h) == calls
    oc = oracle.octree(v, t, vs, threads=2)
    if len(h):
        m = morton_np(h)
        order = np.argsort(m, kind="stable")
        assert np.array_equal(oc["items"], m[order])
        assert oc["aabbs"].tobytes() == vec[order].tobytes()
    # I5
    assert oc["bytes"] == 8 * len(oc["items"]) + 40 * len(oc["nodes"])


def test_regression_vectors():
    with open(os.path.join(GOLD, "regression.json")) as fh:
        G = json.load(fh)
    for key, g in G.items():
        name, vs = key.split("@")
        if name in ("atrium262k",):
            continue  # covered on the GPU box; keeps the CPU suite short
        vs = np.float32(float(vs))
        v, t = vx_scenes.scene(name)
        assert sha(v) == g["verts_sha"] and sha(t) == g["tris_sha"], "scene generator drifted: " + name
        w, calls, gi = oracle.build_bool(v, t, vs)
        a = oracle.bool_aabbs(w, gi, vs)
        assert list(gi["dim"]) == g["dim"] and len(a) == g["occupied"] and calls == g["set_calls"]
        assert sha(w) == g["words_sha"] and sha(a) == g["aabbs_sha"]


def test_octree_structure():
    v, t = vx_scenes.rotated_cube()
    oc = oracle.octree(v, t, 0.05, max_items=16)
    nodes, items = oc["nodes"], oc["items"]
    assert np.all(items[:-1] <= items[1:])
    assert nodes[0]["start"] == 0 and nodes[0]["count"] == len(items)
    leaves = [n for n in nodes if np.all(n["children"] == 0xFFFFFFFF)]
    assert sum(int(n["count"]) for n in leaves) == len(items)
    # root bounds: min = bbox min, max = min + vs * 2^bits
    gi = oracle.grid_info(v, 0.05)
    bits = int(np.ceil(np.log2(max(gi["dim"]))))
    assert np.array_equal(oc["root_min"], gi["bmin"])
    assert np.array_equal(oc["root_max"], gi["bmin"] + np.float32(0.05) * np.float32(1 << bits))


def test_morton_quirk():
    """octTree.hpp:211-218: byte 2 of every coordinate is shifted out of the word."""
    assert oracle.morton3d(1, 0, 0) == 1 and oracle.morton3d(0, 1, 0) == 2 and oracle.morton3d(0, 0, 1) == 4
    assert oracle.morton3d(0xFFFF, 0, 0) == sum(1 << (3 * b) for b in range(16))
    assert oracle.morton3d(0x10000, 0, 0) == 0
    assert oracle.morton3d(0x1FFFF, 5, 9) == oracle.morton3d(0xFFFF, 5, 9)


def test_hit_aabb_formula():
    box = (np.array([0, 0, 0], np.float32), np.array([1, 1, 1], np.float32))
    b = np.zeros(1, dtype=oracle.AABB)
    b["mn"], b["mx"] = box
    assert oracle.hit_aabb(b[0], [-1, 0.5, 0.5], [1, 1e-9, 1e-9]) == pytest.approx(1.0)
    assert oracle.hit_aabb(b[0], [2, 0.5, 0.5], [1, 1e-9, 1e-9]) == -1.0         # behind
    t_inside = oracle.hit_aabb(b[0], [0.5, 0.5, 0.5], [1, 1e-9, 1e-9])
    assert t_inside < 0                                                          # origin inside: t0 < 0 -> not reported (rint:69)
    t, p = oracle.trace_brute(b, np.array([[-1, .5, .5, 1, 1e-9, 1e-9], [0.5, 0.5, 0.5, 1, 1e-9, 1e-9], [-1, .5, .5, -1, 1e-9, 1e-9]], np.float32))
    assert t[0] == pytest.approx(1.0) and p[0] == 0 and t[1] == -1.0 and p[1] == 0xFFFFFFFF and t[2] == -1.0
    # tmin / tmax window (rgen:50-51)
    t, _ = oracle.trace_brute(b, np.array([[-1, .5, .5, 1, 1e-9, 1e-9]], np.float32), tmin=1.5)
    assert t[0] == -1.0
    t, _ = oracle.trace_brute(b, np.array([[-1, .5, .5, 1, 1e-9, 1e-9]], np.float32), tmax=0.5)
    assert t[0] == -1.0


def test_empty_and_flat_inputs():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)  # flat in z: depth 0 -> empty grid
    t = np.array([[0, 1, 2]], np.int32)
    w, calls, gi = oracle.build_bool(v, t, 0.1)
    assert gi["dim"][2] == 0 and w.size == 0 and calls == 0
    oc = oracle.octree(v, t, 0.1)
    assert len(oc["items"]) == 0
    # no triangles
    v, _ = vx_scenes.cube()
    w, calls, gi = oracle.build_bool(v, np.zeros((0, 3), np.int32), 0.25)
    assert not w.any() and calls == 0
    oc = oracle.octree(v, np.zeros((0, 3), np.int32), 0.25)
    assert len(oc["nodes"]) == 0 and len(oc["aabbs"]) == 0


def test_brute_force_inner_form_equals_hit_aabb():
    """The brute-force loop evaluates hitAabb with invDir hoisted out of the box loop and fminf/fmaxf written out; it must be
    the same function, including the NaN products of axis-parallel rays whose origin lies exactly on a box plane
    (inf * 0: rint:49-50) and boxes behind / around the origin."""
    rng = np.random.default_rng(5)
    box = np.zeros(1, dtype=oracle.AABB)[0]
    for k in range(4000):
        mn = rng.uniform(-2, 2, 3).astype(np.float32)
        box["mn"], box["mx"] = mn, mn + rng.uniform(0.01, 1, 3).astype(np.float32)
        o = rng.uniform(-3, 3, 3).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        if k % 4 == 1:
            d[rng.integers(0, 3)] = 0.0
        if k % 4 == 2:
            a = rng.integers(0, 3)
            d[a] = 0.0
            o[a] = box["mn"][a] if k % 8 == 2 else box["mx"][a]
        if k % 4 == 3:
            z = rng.permutation(3)[:2]
            d[z] = 0.0
            o[z[0]] = box["mn"][z[0]]
        a_, b_ = oracle.hit_aabb(box, o, d), oracle.hit_aabb_fast(box, o, d)
        assert (a_ == b_) or (np.isnan(a_) and np.isnan(b_)), (k, a_, b_, box, o, d)


@pytest.mark.parametrize("name,vs", [("cube", 0.25), ("cube", 0.0625), ("rotcube", 0.09), ("adversarial", 0.0625), ("soup2000", 0.02), ("blob70k", 2.0 / 64)])
def test_grid_walk_equals_brute_force(name, vs):
    """oracle/vx_walk.c -- the CPU statement of the traversal the HIP kernel k_walk implements (major-axis slab walk + the exact
    rint formula) and bench.py's CPU stand-in for the ray stage -- returns the brute-force minimum bit for bit: random,
    lattice-corner-aimed, near-axis-parallel, inside-the-grid rays and rays with exactly-zero direction components."""
    from test_gpu_configs import aimed_rays, zero_component_rays
    from test_gpu_parity import axis_rays, corner_rays, inside_rays
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    ow, _, gi = oracle.build_bool(v, t, vs, threads=4)
    oa = oracle.bool_aabbs(ow, gi, vs)
    n = 1500
    for rays in (vx_scenes.random_rays(n, gi["bmin"], gi["bmax"], seed=2), corner_rays(gi, float(vs), n, 5), axis_rays(gi, float(vs), n, 6),
                 inside_rays(gi, float(vs), n, 7), zero_component_rays(oa, gi, vs, 900, 8), aimed_rays(oa, gi, n, 9)):
        ot, op = oracle.trace_brute(oa, rays)
        wt, wp = oracle.trace_walk(ow, gi, vs, rays)
        assert np.array_equal(wt, ot) and np.array_equal(wp, op)
    for tmin, tmax in ((9.5, 10000.0), (0.001, 9.3)):
        rays = vx_scenes.random_rays(n, gi["bmin"], gi["bmax"], seed=3)
        ot, op = oracle.trace_brute(oa, rays, tmin, tmax)
        wt, wp = oracle.trace_walk(ow, gi, vs, rays, tmin, tmax)
        assert np.array_equal(wt, ot) and np.array_equal(wp, op)


def test_grid_walk_far_from_origin():
    """Grids far from the origin: the walk's tolerance scales with max |coordinate| and must still enclose every box the
    brute-force formula can report."""
    from test_gpu_configs import zero_component_rays
    from test_gpu_parity import corner_rays, inside_rays
    rng = np.random.default_rng(77)
    for case in range(12):
        n = int(rng.integers(20, 200))
        scale = float(10.0 ** rng.uniform(-1, 1))
        off = rng.choice([-1.0, 1.0], 3) * rng.uniform(0.0 if case % 2 else 100.0, 2000.0, 3) * scale
        v = (rng.uniform(0, 1, (3 * n, 3)) * scale * rng.uniform(0.05, 1.0, 3) + off).astype(np.float32)
        t = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
        ext = float((v.max(0) - v.min(0)).max())
        vs = np.float32(ext / float(rng.integers(4, 70)))
        ow, _, gi = oracle.build_bool(v, t, vs)
        oa = oracle.bool_aabbs(ow, gi, vs)
        if not len(oa):
            continue
        hi = gi["bmin"] + np.array(gi["dim"], np.float32) * vs
        for rays in (vx_scenes.random_rays(800, gi["bmin"], hi, seed=100 + case), corner_rays(gi, float(vs), 500, case), inside_rays(gi, float(vs), 500, case),
                     zero_component_rays(oa, gi, vs, 300, case)):
            ot, op = oracle.trace_brute(oa, rays)
            wt, wp = oracle.trace_walk(ow, gi, vs, rays)
            assert np.array_equal(wt, ot) and np.array_equal(wp, op), case
