"""ctypes binding for oracle/libvxoracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (libvoxhip.so and its Python/C++ front ends) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VXORACLE_SO: load this build of the oracle instead of the in-tree one (tests/test_oracle_compilers.py compares the
# results of several compilers / optimisation levels; nothing else sets it)
_SO = os.environ.get("VXORACLE_SO") or os.path.join(_HERE, "libvxoracle.so")


def build(force=False):
    if os.environ.get("VXORACLE_SO"):
        return _SO
    srcs = [os.path.join(_HERE, f) for f in ("vx_oracle.c", "vx_walk.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libvxoracle.so"])
    return _SO


class GridInfo(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("center", C.c_float * 3), ("dim", C.c_uint64 * 3)]


AABB = np.dtype([("mn", np.float32, 3), ("mx", np.float32, 3)])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp, ip, u32p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.vxo_grid_info_compute.argtypes = [fp, C.c_size_t, C.c_float, C.POINTER(GridInfo)]
        L.vxo_tri_box_overlap.argtypes = [fp] * 5
        L.vxo_tri_box_overlap.restype = C.c_int
        L.vxo_tri_box_overlap_ss.argtypes = [fp] * 5
        L.vxo_tri_box_overlap_ss.restype = C.c_int
        L.vxo_bool_num_words.argtypes = [C.POINTER(GridInfo)]
        L.vxo_bool_num_words.restype = C.c_uint64
        common = [fp, C.c_size_t, ip, C.c_size_t, C.c_float, C.c_int, C.c_int]
        L.vxo_build_bool.argtypes = common + [u32p]
        L.vxo_build_bool.restype = C.c_uint64
        L.vxo_bool_aabbs.argtypes = [u32p, u64p, C.c_float, fp, C.c_void_p, C.c_uint64]
        L.vxo_bool_aabbs.restype = C.c_uint64
        L.vxo_build_aabbstruct.argtypes = common + [C.c_void_p, C.c_uint64, u64p]
        L.vxo_build_aabbstruct.restype = C.c_uint64
        L.vxo_build_vec.argtypes = common + [C.c_void_p, C.c_uint64]
        L.vxo_build_vec.restype = C.c_uint64
        L.vxo_hits.argtypes = common + [u32p, C.c_uint64]
        L.vxo_hits.restype = C.c_uint64
        L.vxo_morton3d.argtypes = [C.c_uint32] * 3
        L.vxo_morton3d.restype = C.c_uint64
        L.vxo_octree_build.argtypes = [fp, C.c_size_t, ip, C.c_size_t, C.c_float, C.c_uint64, C.c_int]
        L.vxo_octree_build.restype = C.c_void_p
        L.vxo_octree_from_sorted_items.argtypes = [u64p, C.c_uint64, C.c_uint32, C.c_uint64]
        L.vxo_octree_from_sorted_items.restype = C.c_void_p
        for name in ("vxo_octree_num_items", "vxo_octree_num_nodes", "vxo_octree_bytes"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_uint64
        L.vxo_octree_copy_items.argtypes = [C.c_void_p, u64p]
        L.vxo_octree_copy_nodes.argtypes = [C.c_void_p, u32p]
        L.vxo_octree_root.argtypes = [C.c_void_p, fp, fp]
        L.vxo_octree_aabbs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.vxo_octree_aabbs.restype = C.c_uint64
        L.vxo_octree_free.argtypes = [C.c_void_p]
        L.vxo_hit_aabb.argtypes = [C.c_void_p, fp, fp]
        L.vxo_hit_aabb.restype = C.c_float
        L.vxo_hit_aabb_fast.argtypes = [C.c_void_p, fp, fp]
        L.vxo_hit_aabb_fast.restype = C.c_float
        L.vxo_trace_brute.argtypes = [C.c_void_p, C.c_uint64, fp, C.c_uint64, C.c_float, C.c_float, C.c_int, fp, u32p]
        L.vxo_primary_rays.argtypes = [fp, fp, C.c_uint32, C.c_uint32, fp]
        L.vxo_primary_rays_pixels.argtypes = [fp, fp, C.c_uint32, C.c_uint32, u64p, C.c_uint64, fp]
        L.vxo_trace_any_brute.argtypes = [C.c_void_p, C.c_uint64, fp, C.c_uint64, C.c_float, C.c_float, fp, C.POINTER(C.c_uint8)]
        L.vxo_cube_normal.argtypes = [C.c_void_p, fp, fp, C.c_float, fp]
        L.vxo_build_material_ids.argtypes = [fp, C.c_size_t, ip, C.c_size_t, C.c_float, C.c_int, ip, C.c_int32, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64, ip,
                                             C.POINTER(C.c_int32)]
        L.vxo_build_material_ids.restype = C.c_uint64
        L.vxo_shade_image.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, fp, C.c_uint64, fp, fp, C.c_uint32, C.c_uint32, fp, C.c_float, C.c_int, fp, C.c_int, C.c_void_p]
        L.vxo_walk_create.argtypes = [u32p, u64p, C.c_float, fp]
        L.vxo_walk_create.restype = C.c_void_p
        L.vxo_walk_free.argtypes = [C.c_void_p]
        L.vxo_walk_trace.argtypes = [C.c_void_p, fp, C.c_uint64, C.c_float, C.c_float, C.c_int, fp, u64p, u64p]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _u64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _mesh(verts, idx):
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    i = np.ascontiguousarray(idx, dtype=np.int32).reshape(-1, 3)
    return v, i


def grid_info(verts, vs):
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1)
    gi = GridInfo()
    lib().vxo_grid_info_compute(_f(v), v.size, np.float32(vs), C.byref(gi))
    return dict(bmin=np.array(gi.bmin, dtype=np.float32), bmax=np.array(gi.bmax, dtype=np.float32),
                center=np.array(gi.center, dtype=np.float32), dim=tuple(int(d) for d in gi.dim), _c=gi)


def tri_box_overlap(c, h, v0, v1, v2, ss=False):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (c, h, v0, v1, v2)]
    fn = lib().vxo_tri_box_overlap_ss if ss else lib().vxo_tri_box_overlap
    return bool(fn(*[_f(x) for x in a]))


def build_bool(verts, idx, vs, threads=0, sat=-1):
    """-> (words uint32[ceil(N/32)], set_calls, grid_info).  VoxelBuilder<VoxelGridBool,...>::buildVoxelGrid."""
    v, i = _mesh(verts, idx)
    gi = grid_info(v, vs)
    nw = int(lib().vxo_bool_num_words(C.byref(gi["_c"])))
    words = np.zeros(max(nw, 1), dtype=np.uint32)
    calls = lib().vxo_build_bool(_f(v), v.shape[0], _i(i), i.shape[0], np.float32(vs), threads, sat, _u32(words))
    return words[:nw], int(calls), gi


def bool_aabbs(words, gi, vs):
    """VoxelGridBool::getAabbs."""
    dim = np.array(gi["dim"], dtype=np.uint64)
    org = np.ascontiguousarray(gi["bmin"], dtype=np.float32)
    w = np.ascontiguousarray(words, dtype=np.uint32)
    if w.size == 0:
        return np.zeros(0, dtype=AABB)
    n = int(lib().vxo_bool_aabbs(_u32(w), _u64(dim), np.float32(vs), _f(org), None, 0))
    out = np.zeros(n, dtype=AABB)
    if n:
        lib().vxo_bool_aabbs(_u32(w), _u64(dim), np.float32(vs), _f(org), out.ctypes.data, n)
    return out


def build_aabbstruct(verts, idx, vs, threads=0, sat=-1):
    """-> (aabbs, memory_bytes).  VoxelGridAABBstruct path (dense 28 B/voxel on the CPU: keep grids small)."""
    v, i = _mesh(verts, idx)
    by = C.c_uint64(0)
    n = int(lib().vxo_build_aabbstruct(_f(v), v.shape[0], _i(i), i.shape[0], np.float32(vs), threads, sat, None, 0, C.byref(by)))
    out = np.zeros(n, dtype=AABB)
    if n:
        lib().vxo_build_aabbstruct(_f(v), v.shape[0], _i(i), i.shape[0], np.float32(vs), threads, sat, out.ctypes.data, n, C.byref(by))
    return out, int(by.value)


def build_vec(verts, idx, vs, threads=0, sat=-1, cap=None):
    """VoxelGridVec path: one Aabb per setVoxel call, duplicates kept, reference order."""
    v, i = _mesh(verts, idx)
    if cap is None:
        cap = int(lib().vxo_build_vec(_f(v), v.shape[0], _i(i), i.shape[0], np.float32(vs), threads, sat, None, 0))
    out = np.zeros(cap, dtype=AABB)
    n = int(lib().vxo_build_vec(_f(v), v.shape[0], _i(i), i.shape[0], np.float32(vs), threads, sat, out.ctypes.data if cap else None, cap))
    return out[:n]


def hits(verts, idx, vs, threads=0, sat=-1):
    v, i = _mesh(verts, idx)
    n = int(lib().vxo_hits(_f(v), v.shape[0], _i(i), i.shape[0], np.float32(vs), threads, sat, None, 0))
    out = np.zeros((max(n, 1), 3), dtype=np.uint32)
    lib().vxo_hits(_f(v), v.shape[0], _i(i), i.shape[0], np.float32(vs), threads, sat, _u32(out), n)
    return out[:n]


def morton3d(x, y, z):
    return int(lib().vxo_morton3d(int(x), int(y), int(z)))


NODE = np.dtype([("children", np.uint32, 8), ("start", np.uint32), ("count", np.uint32)])


def octree(verts, idx, vs, max_items=16, threads=1):
    """Octree(path, vs, maxItemsPerLeaf) -> dict(items, nodes, aabbs, bytes, root_min, root_max)."""
    v, i = _mesh(verts, idx)
    h = lib().vxo_octree_build(_f(v), v.shape[0], _i(i), i.shape[0], np.float32(vs), max_items, threads)
    if not h:
        raise RuntimeError("We support up to 21 bits per axis (max 2^21 voxels per dimension)!")
    try:
        ni, nn = int(lib().vxo_octree_num_items(h)), int(lib().vxo_octree_num_nodes(h))
        items = np.zeros(max(ni, 1), dtype=np.uint64)
        nodes = np.zeros(max(nn, 1), dtype=NODE)
        lib().vxo_octree_copy_items(h, _u64(items))
        lib().vxo_octree_copy_nodes(h, nodes.ctypes.data_as(C.POINTER(C.c_uint32)))
        aabbs = np.zeros(max(ni, 1), dtype=AABB)
        na = int(lib().vxo_octree_aabbs(h, aabbs.ctypes.data, ni))
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        lib().vxo_octree_root(h, _f(mn), _f(mx))
        return dict(items=items[:ni], nodes=nodes[:nn], aabbs=aabbs[:na], bytes=int(lib().vxo_octree_bytes(h)),
                    root_min=mn, root_max=mx)
    finally:
        lib().vxo_octree_free(h)


def octree_nodes_from_sorted_items(items, bits, max_items=16):
    """Octree::buildTree on an already sorted item list (see vxo_octree_from_sorted_items) -> node array."""
    it = np.ascontiguousarray(items, dtype=np.uint64)
    h = lib().vxo_octree_from_sorted_items(_u64(it), it.size, int(bits), int(max_items))
    if not h:
        raise RuntimeError("octree_nodes_from_sorted_items: more than 21 bits per axis or 2^32 items")
    try:
        nn = int(lib().vxo_octree_num_nodes(h))
        nodes = np.zeros(max(nn, 1), dtype=NODE)
        lib().vxo_octree_copy_nodes(h, nodes.ctypes.data_as(C.POINTER(C.c_uint32)))
        return nodes[:nn]
    finally:
        lib().vxo_octree_free(h)


def morton3d_np(x, y, z):
    """Octree::morton3D (octTree.hpp:211-218, incl. its low-16-bit quirk) on arrays: must equal morton3d element by element."""
    def part(v):
        v = v.astype(np.uint64) & np.uint64(0xFFFF)
        v = (v | (v << np.uint64(16))) & np.uint64(0x0000FF0000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x00F00F00F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0C30C30C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x249249249249)
        return v
    return part(np.asarray(x)) | (part(np.asarray(y)) << np.uint64(1)) | (part(np.asarray(z)) << np.uint64(2))


def hit_aabb(box, o, d):
    b = np.zeros(1, dtype=AABB)
    b[0] = box
    o = np.ascontiguousarray(o, dtype=np.float32)
    d = np.ascontiguousarray(d, dtype=np.float32)
    return float(lib().vxo_hit_aabb(b.ctypes.data, _f(o), _f(d)))


def hit_aabb_fast(box, o, d):
    """The brute-force loop's inlined form of hit_aabb (must equal hit_aabb bit for bit)."""
    b = np.zeros(1, dtype=AABB)
    b[0] = box
    o = np.ascontiguousarray(o, dtype=np.float32)
    d = np.ascontiguousarray(d, dtype=np.float32)
    return float(lib().vxo_hit_aabb_fast(b.ctypes.data, _f(o), _f(d)))


def trace_brute(aabbs, rays, tmin=0.001, tmax=10000.0, threads=None):
    """First hit per ray by brute force over the AABB list -> (t float32[R] (-1 = miss), prim uint32[R])."""
    a = np.ascontiguousarray(aabbs, dtype=AABB)
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    t = np.zeros(r.shape[0], dtype=np.float32)
    p = np.zeros(r.shape[0], dtype=np.uint32)
    if threads is None:
        threads = os.cpu_count() or 1
    lib().vxo_trace_brute(a.ctypes.data if a.size else None, a.size, _f(r), r.shape[0], np.float32(tmin), np.float32(tmax), threads, _f(t), _u32(p))
    return t, p


def primary_rays(view_inv, proj_inv, W, H):
    vi = np.ascontiguousarray(view_inv, dtype=np.float32).reshape(16)
    pi = np.ascontiguousarray(proj_inv, dtype=np.float32).reshape(16)
    rays = np.zeros((H * W, 6), dtype=np.float32)
    lib().vxo_primary_rays(_f(vi), _f(pi), W, H, _f(rays))
    return rays


def primary_rays_pixels(view_inv, proj_inv, W, H, pixels):
    """Primary rays of selected pixels (index py*W + px) of a W x H image."""
    vi = np.ascontiguousarray(view_inv, dtype=np.float32).reshape(16)
    pi = np.ascontiguousarray(proj_inv, dtype=np.float32).reshape(16)
    px = np.ascontiguousarray(pixels, dtype=np.uint64)
    rays = np.zeros((px.size, 6), dtype=np.float32)
    lib().vxo_primary_rays_pixels(_f(vi), _f(pi), W, H, _u64(px), px.size, _f(rays))
    return rays


def trace_any_brute(aabbs, rays, tmin=0.001, tmax=10000.0, tmax_per_ray=None):
    """Shadow query (terminate on first hit): uint8[R], 1 = some box reports an accepted hit."""
    a = np.ascontiguousarray(aabbs, dtype=AABB)
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    out = np.zeros(r.shape[0], dtype=np.uint8)
    tm = None if tmax_per_ray is None else np.ascontiguousarray(tmax_per_ray, dtype=np.float32)
    lib().vxo_trace_any_brute(a.ctypes.data if a.size else None, a.size, _f(r), r.shape[0], np.float32(tmin), np.float32(tmax),
                              _f(tm) if tm is not None else None, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def cube_normals(aabbs, prim, rays, t):
    """raytrace2.rchit:60-73 for every hit (rows of misses are zero)."""
    a = np.ascontiguousarray(aabbs, dtype=AABB)
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    out = np.zeros((r.shape[0], 3), dtype=np.float32)
    tmp = np.zeros(3, dtype=np.float32)
    for i in np.flatnonzero(np.asarray(t) > 0):
        o = np.ascontiguousarray(r[i, :3]); d = np.ascontiguousarray(r[i, 3:])
        lib().vxo_cube_normal(a[int(prim[i]):int(prim[i]) + 1].ctypes.data, _f(o), _f(d), np.float32(t[i]), _f(tmp))
        out[i] = tmp
    return out


def voxel_rank(words, idx):
    """gl_PrimitiveID of voxel index idx: its rank among the set bits of the bitmask (0xFFFFFFFF for ~0 = miss)."""
    w = np.ascontiguousarray(words, dtype=np.uint32)
    idx = np.asarray(idx, dtype=np.uint64)
    pre = np.concatenate([[0], np.cumsum(np.bitwise_count(w).astype(np.uint64))])
    hit = idx != np.uint64(0xFFFFFFFFFFFFFFFF)
    wi = (idx[hit] >> np.uint64(5)).astype(np.int64)
    below = w[wi] & ((np.uint32(1) << (idx[hit] & np.uint64(31)).astype(np.uint32)) - np.uint32(1))
    out = np.full(idx.shape, 0xFFFFFFFF, dtype=np.uint32)
    out[hit] = (pre[wi] + np.bitwise_count(below)).astype(np.uint32)
    return out


def trace_walk(words, gi, vs, rays, tmin=0.001, tmax=10000.0, threads=None, want_stats=False):
    """Grid-walking CPU tracer (vx_walk.c: major-axis slab walk + the exact rint formula) -> (t, prim[, stats]).
    Must equal trace_brute bit for bit; it is the CPU stand-in for the ray stage in bench.py's cpu_baseline."""
    w = np.ascontiguousarray(words, dtype=np.uint32)
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    dim = np.array(gi["dim"], dtype=np.uint64)
    org = np.ascontiguousarray(gi["bmin"], dtype=np.float32)
    t = np.full(r.shape[0], -1.0, dtype=np.float32)
    idx = np.full(r.shape[0], 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    stats = np.zeros(4, dtype=np.uint64)
    if threads is None:
        threads = os.cpu_count() or 1
    if w.size and r.shape[0]:
        h = lib().vxo_walk_create(_u32(w), _u64(dim), np.float32(vs), _f(org))
        try:
            lib().vxo_walk_trace(h, _f(r), r.shape[0], np.float32(tmin), np.float32(tmax), threads, _f(t), _u64(idx), _u64(stats))
        finally:
            lib().vxo_walk_free(h)
    p = voxel_rank(w, idx) if w.size else np.full(r.shape[0], 0xFFFFFFFF, np.uint32)
    if want_stats:
        return t, p, dict(cell_slabs=int(stats[0]), brick_slabs=int(stats[1]), block_slabs=int(stats[2]), exact_tests=int(stats[3]))
    return t, p


def material_ids(verts, idx, vs, tri_value, nvalues, per_call=False, ncalls=0, sat=-1):
    """The reference's commented-out material plumbing (see vx_oracle.c): -> (getMatIdx() int16[], value ids in getMatrials() order)."""
    v, i = _mesh(verts, idx)
    tv = np.ascontiguousarray(tri_value, dtype=np.int32)
    order = np.zeros(max(nvalues, 1), dtype=np.int32)
    used = C.c_int32(0)
    gi = grid_info(v, vs)
    cap = int(ncalls) if per_call else int(np.prod(gi["dim"]))
    out = np.zeros(max(cap, 1), dtype=np.int16)
    n = int(lib().vxo_build_material_ids(_f(v), v.shape[0], _i(i), i.shape[0], np.float32(vs), sat, _i(tv), nvalues, 1 if per_call else 0, int(ncalls),
                                         out.ctypes.data, cap, _i(order), C.byref(used)))
    return out[:n], order[:used.value]


DEFAULT_MATERIAL = np.array([0.1, 0.1, 0.1, 1, 1, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0.1, 0.0, 1.0, 1.0], dtype=np.float32)  # MaterialObj{} (+ illum 0, texture -1)


def shade_image(aabbs, view_inv, proj_inv, W, H, materials=None, mat_idx=None, light_pos=(10.0, 55.0, 8.0), light_intensity=1000.0, light_type=0,
                clear=(1.0, 1.0, 1.0), threads=None):
    """The reference's picture of the voxel boxes (rgen + rint + rchit2 + rmiss + post.frag restated): uint8[H, W, 3].
    materials: records in vx_material layout (80 bytes each) or None = the one default material createAABB uploads."""
    a = np.ascontiguousarray(aabbs, dtype=AABB)
    if materials is None or len(materials) == 0:
        m = np.zeros(20, dtype=np.float32)
        m[:18] = DEFAULT_MATERIAL
        m[18:] = np.array([0, -1], dtype=np.int32).view(np.float32)
        nmat = 1
    else:
        m = np.ascontiguousarray(materials).view(np.float32).reshape(-1)
        nmat = len(materials)
    mi = None if mat_idx is None else np.ascontiguousarray(mat_idx, dtype=np.int16)
    vi = np.ascontiguousarray(view_inv, dtype=np.float32).reshape(16)
    pi = np.ascontiguousarray(proj_inv, dtype=np.float32).reshape(16)
    lp = np.array(light_pos, dtype=np.float32)
    cl = np.array(clear, dtype=np.float32)
    out = np.zeros((H, W, 3), dtype=np.uint8)
    lib().vxo_shade_image(a.ctypes.data if a.size else None, a.size, mi.ctypes.data if mi is not None else None, _f(m), nmat, _f(vi), _f(pi), W, H, _f(lp),
                          np.float32(light_intensity), light_type, _f(cl), threads or (os.cpu_count() or 1), out.ctypes.data)
    return out
