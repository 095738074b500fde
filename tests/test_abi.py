"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/voxhip.h declares, the
host-side logic (OBJ reader, sharding arithmetic, argument checks) behaves, and compute refuses to run without a GPU
instead of falling back to a CPU path."""
import ctypes
import os
import re

import numpy as np
import pytest

import vx_scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "voxhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(vx):
    names = header_functions()
    assert len(names) >= 40
    L = ctypes.CDLL(vx.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libvoxhip.so does not export " + n
    assert sorted(vx.SYMBOLS) == names, "voxhip.py's symbol list and the header disagree"


def test_library_does_not_link_the_oracle(vx):
    import subprocess
    out = subprocess.run(["ldd", vx.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out
    sym = subprocess.run(["nm", "-D", vx.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "vxo_" not in sym


def test_obj_reader_roundtrip(vx, tmp_path):
    v, t = vx_scenes.rotated_cube()
    p = tmp_path / "rot.obj"
    vx_scenes.write_obj(str(p), v, t)
    m = vx.Mesh.load_obj(str(p))
    hv, ht = m.host_arrays()
    assert np.array_equal(hv, v) and np.array_equal(ht, t)      # %.9g round-trips float32 exactly
    # quads, slashes, negative indices, comments
    q = tmp_path / "quad.obj"
    q.write_text("# c\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0 1.0\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4//1 -3//1 -2//1\n")
    m = vx.Mesh.load_obj(str(q))
    hv, ht = m.host_arrays()
    assert hv.shape == (4, 3) and ht.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]


def test_obj_reader_forward_refs_crlf_and_locale(vx, tmp_path):
    """Positive face indices may point at vertices defined further down (resolved after the whole file is read), numbers are
    parsed per line (a short `v` line must not read into the next one, CRLF files), and independently of LC_NUMERIC."""
    p = tmp_path / "fwd.obj"
    p.write_bytes(b"f 1 2 3\r\nv 0.5 1.25\r\nv 1.5 0 0\r\nv +2.5e-1 -3 4\r\nf 1 3 2\r\n")
    hv, ht = vx.Mesh.load_obj(str(p)).host_arrays()
    assert hv.tolist() == [[0.5, 1.25, 0.0], [1.5, 0.0, 0.0], [0.25, -3.0, 4.0]]   # the missing z of vertex 1 stays 0, not 1.5
    assert ht.tolist() == [[0, 1, 2], [0, 2, 1]]
    import locale
    old = locale.setlocale(locale.LC_NUMERIC)
    try:
        for name in ("de_DE.UTF-8", "de_DE", "fr_FR.UTF-8"):
            try:
                locale.setlocale(locale.LC_NUMERIC, name)
                break
            except locale.Error:
                continue
        hv2, _ = vx.Mesh.load_obj(str(p)).host_arrays()
    finally:
        locale.setlocale(locale.LC_NUMERIC, old)
    assert np.array_equal(hv2, hv)
    bad = tmp_path / "oob.obj"
    bad.write_text("f 1 2 4\nv 0 0 0\nv 1 0 0\nv 0 1 0\n")
    with pytest.raises(vx.VxError) as e:
        vx.Mesh.load_obj(str(bad))
    assert e.value.status == 3 and "line 1" in e.value.message


def test_obj_reader_materials(vx, tmp_path):
    """usemtl / mtllib: material id per face and the material records VoxelBuilder would copy (VoxelBuilder.hpp:375-394);
    tinyobj's defaults for fields a .mtl leaves out; unknown names and faces before any usemtl get id -1."""
    (tmp_path / "a.mtl").write_text("newmtl red\nKa 0.1 0.2 0.3\nKd 1 0 0\nKs 0.5 0.5 0.5\nNs 32\nNi 1.45\nd 0.75\nillum 2\nKe 0 0 0.1\n"
                                    "newmtl glass\nKd 0 0 1\nTf 0.9 0.8 0.7\nTr 0.25\n")
    obj = tmp_path / "m.obj"
    obj.write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3\nusemtl glass\nf 1 2 3 4\nusemtl nosuch\nf 1 3 4\nusemtl red\nf 2 3 4\n")
    m = vx.Mesh.load_obj(str(obj))
    recs, ids = m.materials()
    assert ids.tolist() == [-1, 1, 1, -1, 0]
    assert len(recs) == 2
    assert recs[0]["diffuse"].tolist() == [1, 0, 0] and recs[0]["shininess"] == 32 and recs[0]["illum"] == 2
    assert recs[0]["ior"] == np.float32(1.45) and recs[0]["dissolve"] == np.float32(0.75) and recs[0]["texture_id"] == -1
    assert recs[0]["emission"].tolist() == [0, 0, np.float32(0.1)]
    assert recs[1]["transmittance"].tolist() == [np.float32(0.9), np.float32(0.8), np.float32(0.7)] and recs[1]["dissolve"] == np.float32(0.75)
    assert recs[1]["shininess"] == 1 and recs[1]["ior"] == 1 and recs[1]["ambient"].tolist() == [0, 0, 0]   # tinyobj InitMaterial
    # a mesh without mtllib has no materials at all
    v, t = vx_scenes.cube()
    q = tmp_path / "plain.obj"
    vx_scenes.write_obj(str(q), v, t)
    recs, ids = vx.Mesh.load_obj(str(q)).materials()
    assert len(recs) == 0 and ids is None
    # attaching materials to an array mesh
    m2 = vx.Mesh.from_arrays(v, t)
    m2.set_materials(recs := np.zeros(2, vx.MATERIAL), np.arange(len(t), dtype=np.int32) % 2)
    assert m2.materials()[1].tolist() == (np.arange(len(t)) % 2).tolist()
    with pytest.raises(vx.VxError):
        m2.set_materials(recs, np.full(len(t), 2, np.int32))


def test_obj_reader_errors(vx, tmp_path):
    with pytest.raises(vx.VxError) as e:
        vx.Mesh.load_obj(str(tmp_path / "missing.obj"))
    assert e.value.status == 2 and e.value.message == "Path does not exist!"   # VoxelBuilder.hpp:54-56
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(vx.VxError) as e:
        vx.Mesh.load_obj(str(bad))
    assert e.value.status == 3 and e.value.message.startswith("Colud not get valid reader!")  # :63-65


def test_mesh_from_arrays_validates_indices(vx):
    v, t = vx_scenes.cube()
    bad = t.copy()
    bad[3, 1] = 8
    with pytest.raises(vx.VxError):
        vx.Mesh.from_arrays(v, bad)
    bad[3, 1] = -1
    with pytest.raises(vx.VxError):
        vx.Mesh.from_arrays(v, bad)


@pytest.mark.parametrize("nwords,world", [(0, 4), (1, 8), (7, 8), (1000, 3), (4194304, 8), (33554432, 8), (12345677, 6)])
def test_shard_words_partition(vx, nwords, world):
    covered, prev_end = 0, 0
    for r in range(world):
        b, e, pad = vx.shard_words(nwords, r, world)
        assert b == prev_end or b == nwords
        assert b <= e <= nwords and e - b <= pad and pad * world >= nwords
        covered += e - b
        prev_end = e
    assert covered == nwords


def test_shard_range(vx):
    spans = [vx.shard_range(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 9), (9, 10)]


def test_no_cpu_fallback(vx):
    """On a box without a GPU every compute entry point must fail with VX_ERR_NO_DEVICE (never compute on the CPU)."""
    if vx.device_count() > 0:
        pytest.skip("a GPU is present")
    v, t = vx_scenes.cube()
    m = vx.Mesh.from_arrays(v, t)
    with pytest.raises(vx.VxError) as e:
        vx.Grid.voxelize(m, 0.25)
    assert e.value.status == 6
    with pytest.raises(vx.VxError) as e:
        vx.Octree(m, 0.25)
    assert e.value.status == 6
    with pytest.raises(vx.VxError) as e:
        vx.Grid.create(vx.GRID_BOOL, 4, 4, 4, 0.25)
    assert e.value.status == 6


def test_argument_checks(vx):
    v, t = vx_scenes.cube()
    m = vx.Mesh.from_arrays(v, t)
    for bad in (0.0, -1.0, float("nan"), float("inf")):
        with pytest.raises(vx.VxError) as e:
            vx.Grid.voxelize(m, bad)
        assert e.value.status in (1, 6)
