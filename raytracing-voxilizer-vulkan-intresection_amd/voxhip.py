"""ctypes plumbing over libvoxhip.so (the C ABI of include/voxhip.h) for tests, bench.py and __graft_entry__.

This is NOT a second implementation: every call lands in the HIP library; if the library is missing the import
fails loudly.  Nothing here imports or falls back to oracle/.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvoxhip.so")

VX_OK = 0
GRID_BOOL, GRID_AABBSTRUCT, GRID_VEC = 0, 1, 2
VOXELIZE_MATERIALS = 1
VOXELIZE_LIST_ASYNC = 2
STATUS_NAMES = {0: "VX_OK", 1: "VX_ERR_INVALID_ARG", 2: "VX_ERR_PATH", 3: "VX_ERR_PARSE", 4: "VX_ERR_OUT_OF_BOUNDS",
                5: "VX_ERR_MORTON_BITS", 6: "VX_ERR_NO_DEVICE", 7: "VX_ERR_HIP", 8: "VX_ERR_CAPACITY", 9: "VX_ERR_UNSUPPORTED"}

AABB = np.dtype([("mn", np.float32, 3), ("mx", np.float32, 3)])
NODE = np.dtype([("children", np.uint32, 8), ("start", np.uint32), ("count", np.uint32)])
HIT = np.dtype([("ray", np.uint32), ("prim", np.uint32), ("t", np.float32)])
MATERIAL = np.dtype([("ambient", np.float32, 3), ("diffuse", np.float32, 3), ("specular", np.float32, 3), ("transmittance", np.float32, 3),
                     ("emission", np.float32, 3), ("shininess", np.float32), ("ior", np.float32), ("dissolve", np.float32),
                     ("illum", np.int32), ("texture_id", np.int32)])


class GridDesc(C.Structure):
    _fields_ = [("dim", C.c_uint64 * 3), ("voxel_size", C.c_float), ("origin", C.c_float * 3), ("bbox_min", C.c_float * 3),
                ("bbox_max", C.c_float * 3), ("bbox_center", C.c_float * 3), ("num_words", C.c_uint64), ("set_calls", C.c_uint64),
                ("occupied", C.c_uint64), ("triangles", C.c_uint64), ("kind", C.c_int32), ("device", C.c_int32)]


class VoxelizeOpts(C.Structure):
    _fields_ = [("sat_variant", C.c_int32), ("flags", C.c_int32), ("word_begin", C.c_uint64), ("word_end", C.c_uint64),
                ("tri_begin", C.c_uint64), ("tri_end", C.c_uint64), ("stream", C.c_void_p), ("shard_rank", C.c_int32), ("shard_world", C.c_int32)]


class TraceArgs(C.Structure):
    _fields_ = [("rays", C.c_void_p), ("view_inverse", C.POINTER(C.c_float)), ("proj_inverse", C.POINTER(C.c_float)), ("width", C.c_uint32),
                ("height", C.c_uint32), ("num_rays", C.c_uint64), ("tmin", C.c_float), ("tmax", C.c_float), ("tmax_per_ray", C.c_void_p),
                ("any_hit", C.c_int32), ("reserved", C.c_int32), ("t", C.c_void_p), ("prim", C.c_void_p), ("normal", C.c_void_p),
                ("shadowed", C.c_void_p), ("hits", C.c_void_p), ("num_hits", C.c_void_p)]


class VxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), msg))
        self.status = status
        self.message = msg


# every symbol include/voxhip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "vx_last_error", "vx_status_string", "vx_device_count", "vx_set_device", "vx_release_cached_memory",
    "vx_mesh_load_obj", "vx_mesh_from_arrays", "vx_mesh_from_device", "vx_mesh_num_vertices", "vx_mesh_num_triangles",
    "vx_mesh_host_vertices", "vx_mesh_host_indices", "vx_mesh_num_materials", "vx_mesh_materials", "vx_mesh_host_material_ids",
    "vx_mesh_set_materials", "vx_mesh_free",
    "vx_voxelize", "vx_voxelize_into", "vx_voxelize_multi",
    "vx_grid_create", "vx_grid_describe", "vx_grid_set_voxel", "vx_grid_test_voxel", "vx_grid_coords", "vx_grid_bytes",
    "vx_grid_bitmask", "vx_grid_bitmask_device", "vx_grid_bitmask_device_mut", "vx_grid_refresh", "vx_grid_aabbs",
    "vx_grid_aabbs_device", "vx_grid_bind_aabbs_device", "vx_grid_list_wait", "vx_grid_aabbs_device_async", "vx_grid_materials", "vx_grid_material_ids", "vx_grid_material_ids_device", "vx_grid_material_first_use",
    "vx_grid_finish_materials", "vx_multi_create", "vx_multi_voxelize", "vx_multi_grid", "vx_multi_release_grid", "vx_multi_free", "vx_sort_u64", "vx_grid_free",
    "vx_octree_build", "vx_octree_num_items", "vx_octree_num_nodes", "vx_octree_bytes", "vx_octree_items", "vx_octree_nodes",
    "vx_octree_root_bounds", "vx_octree_aabbs", "vx_octree_aabbs_device", "vx_octree_free",
    "vx_trace", "vx_trace_device", "vx_trace_primary_device", "vx_trace_ex", "vx_trace_ex_device",
    "vx_profile_enable", "vx_profile_select", "vx_profile_reset", "vx_profile_read",
    "vx_shard_words", "vx_shard_range",
]

_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's); if
    libvoxhip.so pulled in the system copy first, a later `import torch` would load a second runtime that sees no GPU.
    When torch is installed (tests, bench.py: device memory, streams, torch.distributed), bind to ITS runtime by loading
    it first; plain C/C++ users of libvoxhip.so get the system ROCm runtime via the library's RUNPATH."""
    if os.environ.get("VOXHIP_SYSTEM_HIP_RUNTIME") == "1":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libvoxhip.so is not built (%s): run __graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u64p, fp, u32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.vx_last_error.restype = C.c_char_p
    L.vx_status_string.restype = C.c_char_p
    L.vx_status_string.argtypes = [C.c_int]
    L.vx_device_count.restype = C.c_int
    L.vx_set_device.argtypes = [C.c_int]
    L.vx_mesh_load_obj.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.vx_mesh_from_arrays.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(vp)]
    L.vx_mesh_from_device.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(vp)]
    L.vx_mesh_num_vertices.argtypes = [vp]
    L.vx_mesh_num_vertices.restype = C.c_size_t
    L.vx_mesh_num_triangles.argtypes = [vp]
    L.vx_mesh_num_triangles.restype = C.c_size_t
    L.vx_mesh_host_vertices.argtypes = [vp]
    L.vx_mesh_host_vertices.restype = vp
    L.vx_mesh_host_indices.argtypes = [vp]
    L.vx_mesh_host_indices.restype = vp
    L.vx_mesh_num_materials.argtypes = [vp]
    L.vx_mesh_num_materials.restype = C.c_size_t
    L.vx_mesh_materials.argtypes = [vp, vp, C.c_size_t]
    L.vx_mesh_host_material_ids.argtypes = [vp]
    L.vx_mesh_host_material_ids.restype = vp
    L.vx_mesh_set_materials.argtypes = [vp, vp, C.c_size_t, vp]
    L.vx_mesh_free.argtypes = [vp]
    L.vx_mesh_free.restype = None
    L.vx_voxelize.argtypes = [vp, C.c_float, C.c_int, C.POINTER(VoxelizeOpts), C.POINTER(vp)]
    L.vx_voxelize_into.argtypes = [vp, C.c_float, C.POINTER(VoxelizeOpts), vp]
    L.vx_voxelize_multi.argtypes = [vp, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(vp)]
    L.vx_sort_u64.argtypes = [vp, C.c_uint64, C.c_int]
    L.vx_multi_create.argtypes = [vp, C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(vp)]
    L.vx_multi_voxelize.argtypes = [vp, C.c_float, vp, C.c_int]
    L.vx_multi_grid.argtypes = [vp, C.c_int]
    L.vx_multi_grid.restype = vp
    L.vx_multi_release_grid.argtypes = [vp, C.c_int]
    L.vx_multi_release_grid.restype = vp
    L.vx_multi_free.argtypes = [vp]
    L.vx_multi_free.restype = None
    L.vx_grid_material_first_use.argtypes = [vp, C.POINTER(C.c_int64), C.c_uint64, C.POINTER(C.c_uint64)]
    L.vx_grid_finish_materials.argtypes = [vp, C.POINTER(C.c_int64), C.c_uint64]
    L.vx_grid_create.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, fp, vp, C.POINTER(vp)]
    L.vx_grid_describe.argtypes = [vp, C.POINTER(GridDesc)]
    L.vx_grid_set_voxel.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64]
    L.vx_grid_test_voxel.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
    L.vx_grid_coords.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, fp]
    L.vx_grid_bytes.argtypes = [vp]
    L.vx_grid_bytes.restype = C.c_uint64
    L.vx_grid_bitmask.argtypes = [vp, vp, C.c_uint64]
    L.vx_grid_bitmask_device.argtypes = [vp]
    L.vx_grid_bitmask_device.restype = vp
    L.vx_grid_bitmask_device_mut.argtypes = [vp]
    L.vx_grid_bitmask_device_mut.restype = vp
    L.vx_grid_refresh.argtypes = [vp]
    L.vx_grid_aabbs.argtypes = [vp, vp, C.c_uint64, u64p]
    L.vx_grid_aabbs_device.argtypes = [vp, vp, C.c_uint64, u64p]
    L.vx_grid_bind_aabbs_device.argtypes = [vp, vp, C.c_uint64]
    L.vx_grid_list_wait.argtypes = [vp]
    L.vx_grid_aabbs_device_async.argtypes = [vp, vp, C.c_uint64, u64p]
    L.vx_grid_materials.argtypes = [vp, vp, C.c_uint64, u64p]
    L.vx_grid_material_ids.argtypes = [vp, vp, C.c_uint64, u64p]
    L.vx_grid_material_ids_device.argtypes = [vp]
    L.vx_grid_material_ids_device.restype = vp
    L.vx_grid_free.argtypes = [vp]
    L.vx_grid_free.restype = None
    L.vx_octree_build.argtypes = [vp, C.c_float, C.c_uint64, vp, C.POINTER(vp)]
    for n in ("vx_octree_num_items", "vx_octree_num_nodes", "vx_octree_bytes"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = C.c_uint64
    L.vx_octree_items.argtypes = [vp, vp, C.c_uint64]
    L.vx_octree_nodes.argtypes = [vp, vp, C.c_uint64]
    L.vx_octree_root_bounds.argtypes = [vp, fp, fp]
    L.vx_octree_aabbs.argtypes = [vp, vp, C.c_uint64, u64p]
    L.vx_octree_aabbs_device.argtypes = [vp, vp, C.c_uint64, u64p]
    L.vx_octree_free.argtypes = [vp]
    L.vx_octree_free.restype = None
    L.vx_trace.argtypes = [vp, vp, C.c_uint64, C.c_float, C.c_float, vp, vp, u64p]
    L.vx_trace_device.argtypes = [vp, vp, C.c_uint64, C.c_float, C.c_float, vp, vp, vp, vp]
    L.vx_trace_primary_device.argtypes = [vp, fp, fp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, vp, vp]
    L.vx_profile_enable.argtypes = [C.c_int]
    L.vx_profile_select.argtypes = [C.c_char_p]
    L.vx_profile_read.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), u64p]
    L.vx_trace_ex.argtypes = [vp, C.POINTER(TraceArgs)]
    L.vx_trace_ex_device.argtypes = [vp, C.POINTER(TraceArgs)]
    L.vx_shard_words.argtypes = [C.c_uint64, C.c_int, C.c_int, u64p, u64p, u64p]
    L.vx_shard_words.restype = None
    L.vx_shard_range.argtypes = [C.c_uint64, C.c_int, C.c_int, u64p, u64p]
    L.vx_shard_range.restype = None
    _lib = L
    return L


def _check(status):
    if status != VX_OK:
        raise VxError(status, lib().vx_last_error().decode("utf-8", "replace"))


def device_count():
    return lib().vx_device_count()


def set_device(d):
    _check(lib().vx_set_device(d))


def profile_enable(on=True):
    lib().vx_profile_enable(1 if on else 0)


def profile_select(kernel=None):
    """Time only this kernel (bare name); None = all kernels."""
    lib().vx_profile_select(kernel.encode() if kernel else None)


def profile_reset():
    lib().vx_profile_reset()


def profile_read():
    """{kernel base name: (total_ms, launches)} accumulated since the last reset (template arguments folded)."""
    out = {}
    slot = 0
    while True:
        name = C.create_string_buffer(128)
        ms, n = C.c_double(), C.c_uint64()
        if lib().vx_profile_read(slot, name, 128, C.byref(ms), C.byref(n)) != VX_OK:
            break
        key = name.value.decode().lstrip("(").split("<")[0].rstrip(")")
        a, b = out.get(key, (0.0, 0))
        out[key] = (a + ms.value, b + n.value)
        slot += 1
    return out


def shard_words(num_words, rank, world):
    b, e, p = C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib().vx_shard_words(num_words, rank, world, C.byref(b), C.byref(e), C.byref(p))
    return b.value, e.value, p.value


def shard_range(count, rank, world):
    b, e = C.c_uint64(), C.c_uint64()
    lib().vx_shard_range(count, rank, world, C.byref(b), C.byref(e))
    return b.value, e.value


class Mesh:
    """vx_mesh handle (what VoxelBuilder keeps after readObjFile)."""

    def __init__(self, handle, keep=None):
        self.h = handle
        self._keep = keep

    @classmethod
    def load_obj(cls, path):
        h = C.c_void_p()
        _check(lib().vx_mesh_load_obj(os.fsencode(path), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, verts, tris):
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        h = C.c_void_p()
        _check(lib().vx_mesh_from_arrays(v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0], C.byref(h)))
        return cls(h)

    @classmethod
    def from_device(cls, verts_ptr, nverts, tris_ptr, ntris, keep=None):
        h = C.c_void_p()
        _check(lib().vx_mesh_from_device(verts_ptr, nverts, tris_ptr, ntris, C.byref(h)))
        return cls(h, keep)

    @property
    def num_vertices(self):
        return lib().vx_mesh_num_vertices(self.h)

    @property
    def num_triangles(self):
        return lib().vx_mesh_num_triangles(self.h)

    def host_arrays(self):
        nv, nt = self.num_vertices, self.num_triangles
        vp_, ip_ = lib().vx_mesh_host_vertices(self.h), lib().vx_mesh_host_indices(self.h)
        v = np.ctypeslib.as_array(C.cast(vp_, C.POINTER(C.c_float)), shape=(nv * 3,)).reshape(nv, 3).copy() if nv else np.zeros((0, 3), np.float32)
        t = np.ctypeslib.as_array(C.cast(ip_, C.POINTER(C.c_int32)), shape=(nt * 3,)).reshape(nt, 3).copy() if nt else np.zeros((0, 3), np.int32)
        return v, t

    def materials(self):
        """(records MATERIAL[n], per-triangle ids int32[T] or None): tinyobj's GetMaterials() / mesh.material_ids."""
        n = lib().vx_mesh_num_materials(self.h)
        recs = np.zeros(n, dtype=MATERIAL)
        if n:
            _check(lib().vx_mesh_materials(self.h, recs.ctypes.data, n))
        ip_ = lib().vx_mesh_host_material_ids(self.h)
        nt = self.num_triangles
        ids = np.ctypeslib.as_array(C.cast(ip_, C.POINTER(C.c_int32)), shape=(nt,)).copy() if (ip_ and nt) else None
        return recs, ids

    def set_materials(self, records, tri_ids):
        r = np.ascontiguousarray(records, dtype=MATERIAL)
        ids = None if tri_ids is None else np.ascontiguousarray(tri_ids, dtype=np.int32)
        _check(lib().vx_mesh_set_materials(self.h, r.ctypes.data if r.size else None, r.size, ids.ctypes.data if ids is not None else None))

    def free(self):
        if self.h:
            lib().vx_mesh_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Grid:
    """vx_grid handle (a VoxelGridBool / VoxelGridAABBstruct / VoxelGridVec)."""
    owned = True


    def __init__(self, handle):
        self.h = handle

    @classmethod
    def voxelize(cls, mesh, voxel_size, kind=GRID_BOOL, sat_variant=0, words=None, tris=None, stream=None, materials=False, shard=None):
        o = VoxelizeOpts()
        o.sat_variant = sat_variant
        o.flags = VOXELIZE_MATERIALS if materials else 0
        if shard is not None:
            o.shard_rank, o.shard_world = shard
        if words is not None:
            o.word_begin, o.word_end = words
        if tris is not None:
            o.tri_begin, o.tri_end = tris
        o.stream = stream
        h = C.c_void_p()
        _check(lib().vx_voxelize(mesh.h, np.float32(voxel_size), kind, C.byref(o), C.byref(h)))
        return cls(h)

    @classmethod
    def voxelize_multi(cls, mesh, voxel_size, devices, kind=GRID_BOOL, sat_variant=0, all_gather=False):
        """vx_voxelize_multi: word shards on the given devices + peer copies -> [grid on devices[0]] or one grid per device."""
        dv = (C.c_int * len(devices))(*devices)
        n = len(devices) if all_gather else 1
        hs = (C.c_void_p * n)()
        _check(lib().vx_voxelize_multi(mesh.h, np.float32(voxel_size), kind, sat_variant, dv, len(devices), 1 if all_gather else 0, hs))
        return [cls(C.c_void_p(hs[i])) for i in range(n)]

    def revoxelize(self, mesh, voxel_size, sat_variant=0, words=None, tris=None, stream=None, materials=False, shard=None, list_async=False):
        o = VoxelizeOpts()
        o.sat_variant = sat_variant
        o.flags = (VOXELIZE_MATERIALS if materials else 0) | (VOXELIZE_LIST_ASYNC if list_async else 0)
        if shard is not None:
            o.shard_rank, o.shard_world = shard
        if words is not None:
            o.word_begin, o.word_end = words
        if tris is not None:
            o.tri_begin, o.tri_end = tris
        o.stream = stream
        _check(lib().vx_voxelize_into(mesh.h, np.float32(voxel_size), C.byref(o), self.h))

    @classmethod
    def create(cls, kind, x, y, z, voxel_size, origin=(0.0, 0.0, 0.0), stream=None):
        org = (C.c_float * 3)(*origin)
        h = C.c_void_p()
        _check(lib().vx_grid_create(kind, x, y, z, np.float32(voxel_size), org, stream, C.byref(h)))
        return cls(h)

    def describe(self):
        d = GridDesc()
        _check(lib().vx_grid_describe(self.h, C.byref(d)))
        return dict(dim=tuple(int(x) for x in d.dim), voxel_size=float(d.voxel_size), origin=np.array(d.origin, np.float32),
                    bbox_min=np.array(d.bbox_min, np.float32), bbox_max=np.array(d.bbox_max, np.float32),
                    bbox_center=np.array(d.bbox_center, np.float32), num_words=int(d.num_words), set_calls=int(d.set_calls),
                    occupied=int(d.occupied), triangles=int(d.triangles), kind=int(d.kind), device=int(d.device))

    def set_voxel(self, x, y, z):
        _check(lib().vx_grid_set_voxel(self.h, x, y, z))

    def test_voxel(self, x, y, z):
        r = C.c_int()
        _check(lib().vx_grid_test_voxel(self.h, x, y, z, C.byref(r)))
        return bool(r.value)

    def coords(self, x, y, z):
        o = (C.c_float * 3)()
        _check(lib().vx_grid_coords(self.h, x, y, z, o))
        return np.array(o, dtype=np.float32)

    def memory_bytes(self):
        return int(lib().vx_grid_bytes(self.h))

    def bitmask(self):
        n = self.describe()["num_words"]
        w = np.zeros(max(n, 1), dtype=np.uint32)
        _check(lib().vx_grid_bitmask(self.h, w.ctypes.data, n))
        return w[:n]

    def bitmask_device_ptr(self, mutable=False):
        return lib().vx_grid_bitmask_device_mut(self.h) if mutable else lib().vx_grid_bitmask_device(self.h)

    def refresh(self):
        _check(lib().vx_grid_refresh(self.h))

    def aabbs(self):
        n = C.c_uint64()
        _check(lib().vx_grid_aabbs(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=AABB)
        if n.value:
            _check(lib().vx_grid_aabbs(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def material_first_use(self):
        """Sharded VX_VOXELIZE_MATERIALS build: per material value the first triangle of this shard that uses it (-1: none)."""
        n = C.c_uint64()
        _check(lib().vx_grid_material_first_use(self.h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int64)
        _check(lib().vx_grid_material_first_use(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)), n.value, None))
        return out[:n.value]

    def finish_materials(self, first_use_min):
        fu = np.ascontiguousarray(first_use_min, dtype=np.int64)
        _check(lib().vx_grid_finish_materials(self.h, fu.ctypes.data_as(C.POINTER(C.c_int64)), fu.size))

    def materials(self):
        """(getMatrials() as MATERIAL records, getMatIdx() as int16[]) -- empty unless built with materials=True."""
        n = C.c_uint64()
        _check(lib().vx_grid_materials(self.h, None, 0, C.byref(n)))
        recs = np.zeros(n.value, dtype=MATERIAL)
        if n.value:
            _check(lib().vx_grid_materials(self.h, recs.ctypes.data, n.value, C.byref(n)))
        _check(lib().vx_grid_material_ids(self.h, None, 0, C.byref(n)))
        ids = np.zeros(n.value, dtype=np.int16)
        if n.value:
            _check(lib().vx_grid_material_ids(self.h, ids.ctypes.data, n.value, C.byref(n)))
        return recs, ids

    def bind_aabbs_device(self, dev_ptr, capacity):
        """VX_GRID_VEC: later revoxelize() calls build the list straight in this device buffer (None / 0 removes the binding)."""
        _check(lib().vx_grid_bind_aabbs_device(self.h, dev_ptr, capacity))

    def list_wait(self):
        """VX_VOXELIZE_LIST_ASYNC builds: work queued on the grid's stream after this call sees the complete list."""
        _check(lib().vx_grid_list_wait(self.h))

    def aabbs_device_async(self, dev_ptr, capacity):
        """vx_grid_aabbs_device_async: the count now, the records beside the next ray batch (or after list_wait())."""
        n = C.c_uint64()
        _check(lib().vx_grid_aabbs_device_async(self.h, dev_ptr, capacity, C.byref(n)))
        return n.value

    def aabbs_device(self, dev_ptr, capacity):
        n = C.c_uint64()
        _check(lib().vx_grid_aabbs_device(self.h, dev_ptr, capacity, C.byref(n)))
        return n.value

    def trace(self, rays, tmin=0.001, tmax=10000.0, want_prim=True):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        t = np.zeros(r.shape[0], dtype=np.float32)
        p = np.zeros(r.shape[0], dtype=np.uint32) if want_prim else None
        nh = C.c_uint64()
        _check(lib().vx_trace(self.h, r.ctypes.data, r.shape[0], np.float32(tmin), np.float32(tmax), t.ctypes.data,
                              p.ctypes.data if want_prim else None, C.byref(nh)))
        return (t, p, nh.value) if want_prim else (t, nh.value)

    def trace_ex(self, rays=None, camera=None, tmin=0.001, tmax=10000.0, tmax_per_ray=None, any_hit=False, want=("t", "prim")):
        """Host-buffer extended query -> dict of the requested outputs (t, prim, normal, shadowed)."""
        a = TraceArgs()
        keep = []
        if rays is not None:
            r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
            keep.append(r)
            a.rays, a.num_rays, n = r.ctypes.data, r.shape[0], r.shape[0]
        else:
            vi, pi, w, h = camera
            cvi = (C.c_float * 16)(*[float(x) for x in np.asarray(vi).reshape(16)])
            cpi = (C.c_float * 16)(*[float(x) for x in np.asarray(pi).reshape(16)])
            keep += [cvi, cpi]
            a.view_inverse, a.proj_inverse, a.width, a.height, n = cvi, cpi, w, h, w * h
        a.tmin, a.tmax, a.any_hit = np.float32(tmin), np.float32(tmax), 1 if any_hit else 0
        if tmax_per_ray is not None:
            tm = np.ascontiguousarray(tmax_per_ray, dtype=np.float32)
            keep.append(tm)
            a.tmax_per_ray = tm.ctypes.data
        out = {}
        if "t" in want:
            out["t"] = np.zeros(n, np.float32); a.t = out["t"].ctypes.data
        if "prim" in want:
            out["prim"] = np.zeros(n, np.uint32); a.prim = out["prim"].ctypes.data
        if "normal" in want:
            out["normal"] = np.zeros((n, 3), np.float32); a.normal = out["normal"].ctypes.data
        if "shadowed" in want:
            out["shadowed"] = np.zeros(n, np.uint8); a.shadowed = out["shadowed"].ctypes.data
        _check(lib().vx_trace_ex(self.h, C.byref(a)))
        return out

    def trace_device(self, rays_ptr, nrays, t_ptr, prim_ptr=None, hits_ptr=None, nhits_ptr=None, tmin=0.001, tmax=10000.0):
        _check(lib().vx_trace_device(self.h, rays_ptr, nrays, np.float32(tmin), np.float32(tmax), t_ptr, prim_ptr, hits_ptr, nhits_ptr))

    def trace_primary_device(self, view_inv, proj_inv, width, height, t_ptr, prim_ptr=None, tmin=0.001, tmax=10000.0):
        vi = (C.c_float * 16)(*[float(x) for x in np.asarray(view_inv).reshape(16)])
        pi = (C.c_float * 16)(*[float(x) for x in np.asarray(proj_inv).reshape(16)])
        _check(lib().vx_trace_primary_device(self.h, vi, pi, width, height, np.float32(tmin), np.float32(tmax), t_ptr, prim_ptr))

    def free(self):
        if self.h and self.owned:
            lib().vx_grid_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sort_u64(keys, bits):
    """vx_sort_u64: the octree's device radix sort on a host array (returns a sorted copy)."""
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    _check(lib().vx_sort_u64(k.ctypes.data, k.size, int(bits)))
    return k


class BorrowedGrid(Grid):
    """A grid owned by a vx_multi context (valid until its next voxelize / free)."""
    owned = False


class Multi:
    """vx_multi handle: a mesh resident on several devices (logical ranks allowed), one grid and one worker thread per rank;
    voxelize() rebuilds the grid in steady state (word shards by rank + peer copies)."""

    def __init__(self, mesh, devices, kind=GRID_BOOL):
        dv = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(lib().vx_multi_create(mesh.h, dv, len(devices), kind, C.byref(h)))
        self.h = h
        self.n = len(devices)

    def voxelize(self, voxel_size, sat_variant=0, materials=False, all_gather=False):
        o = VoxelizeOpts()
        o.sat_variant = sat_variant
        o.flags = VOXELIZE_MATERIALS if materials else 0
        _check(lib().vx_multi_voxelize(self.h, np.float32(voxel_size), C.byref(o), 1 if all_gather else 0))
        return [BorrowedGrid(C.c_void_p(lib().vx_multi_grid(self.h, k))) for k in range(self.n if all_gather else 1)]

    def free(self):
        if self.h:
            lib().vx_multi_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Octree:
    """vx_octree handle (the reference's Octree)."""

    def __init__(self, mesh, voxel_size, max_items=16, stream=None):
        h = C.c_void_p()
        _check(lib().vx_octree_build(mesh.h, np.float32(voxel_size), max_items, stream, C.byref(h)))
        self.h = h

    @property
    def num_items(self):
        return int(lib().vx_octree_num_items(self.h))

    @property
    def num_nodes(self):
        return int(lib().vx_octree_num_nodes(self.h))

    def memory_bytes(self):
        return int(lib().vx_octree_bytes(self.h))

    def items(self):
        n = self.num_items
        out = np.zeros(max(n, 1), dtype=np.uint64)
        _check(lib().vx_octree_items(self.h, out.ctypes.data, n))
        return out[:n]

    def nodes(self):
        n = self.num_nodes
        out = np.zeros(max(n, 1), dtype=NODE)
        _check(lib().vx_octree_nodes(self.h, out.ctypes.data, n))
        return out[:n]

    def root_bounds(self):
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        _check(lib().vx_octree_root_bounds(self.h, mn, mx))
        return np.array(mn, np.float32), np.array(mx, np.float32)

    def aabbs(self):
        n = C.c_uint64()
        _check(lib().vx_octree_aabbs(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=AABB)
        if n.value:
            _check(lib().vx_octree_aabbs(self.h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def free(self):
        if self.h:
            lib().vx_octree_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
