"""GPU tests of the drop-in C++ surface: the facade classes (VoxelBuilder<T,bool>, VoxelGridBool/AABBstruct/Vec, Octree)
and the `voxilizer <obj> <voxelsize>` CLI, compiled with plain g++ against libvoxhip.so, compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle
import vx_scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")


def run(cmd):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = PKG + ":" + env.get("LD_LIBRARY_PATH", "")
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)


@pytest.mark.parametrize("name,vs", [("rotcube", 0.09), ("adversarial", 0.0625), ("blob70k", 2.0 / 64)])
def test_facade_classes(gpu, tmp_path, name, vs):
    v, t = vx_scenes.scene(name)
    obj = tmp_path / (name + ".obj")
    vx_scenes.write_obj(str(obj), v, t)
    out = tmp_path / "out"
    out.mkdir()
    r = run([os.path.join(PKG, "facade_selftest"), str(obj), repr(float(np.float32(vs))), str(out)])
    assert r.returncode == 0 and "SELFTEST OK" in r.stdout, r.stdout
    vs = np.float32(vs)
    ow, calls, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    assert open(out / "bool.words", "rb").read() == ow.tobytes()
    assert open(out / "bool.aabb", "rb").read() == oa.tobytes()
    ow2, _, _ = oracle.build_bool(v, t, vs, threads=2)           # VoxelBuilder<T,true>: threaded driver's SAT
    assert open(out / "bool_parallel.aabb", "rb").read() == oracle.bool_aabbs(ow2, gi, vs).tobytes()
    assert open(out / "aabbstruct.aabb", "rb").read() == oa.tobytes()
    assert open(out / "vec.aabb", "rb").read() == oracle.build_vec(v, t, vs).tobytes()
    oc = oracle.octree(v, t, vs, threads=2)
    assert open(out / "octree.aabb", "rb").read() == oc["aabbs"].tobytes()
    assert open(out / "octree.nodes", "rb").read() == oc["nodes"].tobytes()
    # the reference's stdout lines (VoxelBuilder.hpp:343-352,417)
    assert "Grid dimensions: %dx%dx%d" % gi["dim"] in r.stdout
    assert "Total triangles processed: %d" % len(t) in r.stdout
    assert "Bounding box: min(" in r.stdout and "Voxel size: " in r.stdout


def test_cli_contract(gpu, tmp_path):
    """`<Path to obj file> <Voxlesize>` (README.md:57, main.cpp:163) -> createAABB's three result lines
    (hello_vulkan.cpp:686-688) and, with --dump, the AABB list the renderer would upload."""
    v, t = vx_scenes.cube()
    obj = tmp_path / "cube.obj"
    vx_scenes.write_obj(str(obj), v, t)
    exe = os.path.join(PKG, "voxilizer")
    dump = tmp_path / "a.bin"
    r = run([exe, str(obj), "0.0625", "--dump", str(dump)])
    assert r.returncode == 0, r.stdout
    for line in ("Voxel build took ", "Aabb build took ", "Total usage of the VoxelGridAABBstruct is 4096", "Grid dimensions: 32x32x32",
                 "Total triangles processed: 12"):
        assert line in r.stdout, r.stdout
    ow, _, gi = oracle.build_bool(v, t, np.float32(0.0625))
    assert open(dump, "rb").read() == oracle.bool_aabbs(ow, gi, np.float32(0.0625)).tobytes()
    # BASELINE configs[0] literally: cube.obj at voxelsize 0.1 -> a 20^3 grid with ZERO occupied voxels (knife edge)
    r = run([exe, str(obj), "0.1", "--dump", str(dump)])
    assert r.returncode == 0 and "Grid dimensions: 20x20x20" in r.stdout and os.path.getsize(dump) == 0
    # octree flavour and Benchmaker printout
    r = run([exe, str(obj), "0.25", "--grid", "octree"])
    assert r.returncode == 0 and "Total usage of the Octree is 4144" in r.stdout, r.stdout
    r = run([exe, str(obj), "0.25", "--bench", "3"])
    assert r.returncode == 0 and "Voxel build took on avrage" in r.stdout and "AABB build took on avrage" in r.stdout
    # the build spread over logical ranks (vx_voxelize_multi): the same list
    r = run([exe, str(obj), "0.0625", "--dump", str(dump), "--gpus", "4", "--logical"])
    assert r.returncode == 0 and "build sharded over 4 ranks" in r.stdout, r.stdout
    assert open(dump, "rb").read() == oracle.bool_aabbs(ow, gi, np.float32(0.0625)).tobytes()
    r = run([exe, str(obj), "0.0625", "--gpus", "64"])
    assert r.returncode == 2 and "device(s) visible" in r.stdout
    # error paths
    r = run([exe, str(tmp_path / "nope.obj"), "0.1"])
    assert r.returncode == 1 and "Path does not exist!" in r.stdout
    r = run([exe])
    assert r.returncode == 2


def test_cli_render(gpu, tmp_path):
    """--render: the reference's picture of the voxels without Vulkan.  Checks the PPM container, that the
    number of non-background pixels equals the number of primary-ray hits the library reports for the same camera, and that
    both lit and shadowed/back-facing shades occur."""
    v, t = vx_scenes.rotated_cube(half=1.0, offset=(0.0, 1.0, 0.0))
    obj = tmp_path / "c.obj"
    vx_scenes.write_obj(str(obj), v, t)
    ppm = tmp_path / "o.ppm"
    r = run([os.path.join(PKG, "voxilizer"), str(obj), "0.05", "--render", str(ppm), "--size", "320x180"])
    assert r.returncode == 0 and "rendered 320x180" in r.stdout, r.stdout
    raw = open(ppm, "rb").read()
    assert raw.startswith(b"P6\n320 180\n255\n")
    img = np.frombuffer(raw[len(b"P6\n320 180\n255\n"):], np.uint8).reshape(180, 320, 3)
    bg = int(round((0.8 ** (1 / 2.2)) * 255))
    is_bg = np.all(img == bg, axis=2)
    g = gpu.Grid.voxelize(gpu.Mesh.from_arrays(v, t), np.float32(0.05))
    vi, pi = vx_scenes.camera_matrices(aspect=320.0 / 180.0)
    tt = g.trace_ex(camera=(vi, pi, 320, 180), want=("t",))["t"].reshape(180, 320)
    assert abs(int((~is_bg).sum()) - int((tt > 0).sum())) <= 2     # the two camera set-ups differ in the last float bits
    shades = np.unique(img[~is_bg][:, 0])
    assert len(shades) >= 2 and img[~is_bg][:, 2].max() == 0        # yellow default material: blue channel stays 0


def test_cli_render_matches_oracle_shade(gpu, tmp_path):
    """--render (+ --materials) per pixel against the oracle's restatement of the reference's shaders (rgen, rint, rchit2, rmiss,
    post.frag) on the oracle's own boxes and material ids: within 1 LSB, except where a silhouette or shadow edge falls between
    the two ray set-ups' last float bits (a handful of pixels)."""
    v, t = vx_scenes.rotated_cube(half=1.0, offset=(0.0, 1.0, 0.0))
    (tmp_path / "m.mtl").write_text("newmtl shiny\nKa 0.05 0.05 0.1\nKd 0.2 0.4 0.9\nKs 0.9 0.9 0.9\nNs 48\nillum 2\n"
                                    "newmtl matte\nKa 0.1 0.0 0.0\nKd 0.9 0.3 0.2\nillum 1\n")
    obj = tmp_path / "c.obj"
    lines = ["mtllib m.mtl"] + ["v %.9g %.9g %.9g" % tuple(p) for p in v.tolist()]
    ids = []
    for k, tri in enumerate(t.tolist()):
        mid = [-1, 0, 1][(k // 2) % 3]
        lines.append({-1: "usemtl none_such", 0: "usemtl shiny", 1: "usemtl matte"}[mid])
        lines.append("f %d %d %d" % (tri[0] + 1, tri[1] + 1, tri[2] + 1))
        ids.append(mid)
    obj.write_text("\n".join(lines) + "\n")
    W, H = 240, 135
    vs = np.float32(0.05)
    ow, _, gi = oracle.build_bool(v, t, vs)
    oa = oracle.bool_aabbs(ow, gi, vs)
    for with_mat in (False, True, "sharded"):
        ppm, cam, md = tmp_path / "o.ppm", tmp_path / "cam.bin", tmp_path / "mat.bin"
        cmd = [os.path.join(PKG, "voxilizer"), str(obj), "0.05", "--render", str(ppm), "--size", "%dx%d" % (W, H), "--camera-dump", str(cam)]
        if with_mat:
            cmd += ["--materials", "--dump-materials", str(md)]
        if with_mat == "sharded":   # VoxelBuilder::withMaterials() + withDevices(): the same ids from three word shards (logical ranks)
            cmd += ["--gpus", "3", "--logical"]
        r = run(cmd)
        assert r.returncode == 0, r.stdout
        raw = open(ppm, "rb").read()
        hdr = b"P6\n%d %d\n255\n" % (W, H)
        assert raw.startswith(hdr)
        img = np.frombuffer(raw[len(hdr):], np.uint8).reshape(H, W, 3)
        cm = np.fromfile(cam, np.float32)
        vi, pi = cm[:16], cm[16:]
        mats, midx = None, None
        if with_mat:
            # value ids as the library forms them: 0 = MaterialObj{}, 1 = shiny, 2 = matte; faces with an unknown usemtl carry the default
            tv = np.array([0 if i < 0 else i + 1 for i in ids], np.int32)
            midx, order = oracle.material_ids(v, t, vs, tv, 3)
            assert np.array_equal(np.fromfile(md, np.int16), midx)
            recs = np.zeros(3, dtype=gpu.MATERIAL)
            recs[0] = (tuple([0.1] * 3), (1, 1, 0), (1, 1, 1), (0, 0, 0), (0, 0, 0.1), 0, 1, 1, 0, -1)
            recs[1] = ((0.05, 0.05, 0.1), (0.2, 0.4, 0.9), (0.9, 0.9, 0.9), (0, 0, 0), (0, 0, 0), 48, 1, 1, 2, -1)
            recs[2] = ((0.1, 0, 0), (0.9, 0.3, 0.2), (0, 0, 0), (0, 0, 0), (0, 0, 0), 1, 1, 1, 1, -1)
            mats = recs[order]
        exp = oracle.shade_image(oa, vi, pi, W, H, materials=mats, mat_idx=midx)
        diff = np.abs(img.astype(np.int16) - exp.astype(np.int16)).max(axis=2)
        assert (diff <= 1).mean() > 0.997, "materials=%s: %d of %d pixels differ by more than 1 LSB" % (with_mat, int((diff > 1).sum()), W * H)
        bg = int(round((0.8 ** (1 / 2.2)) * 255))
        hit = ~np.all(exp == bg, axis=2)
        assert 0.02 < hit.mean() < 0.9
        if with_mat:
            assert len(np.unique(exp[hit].reshape(-1, 3), axis=0)) > 8      # several materials, lit / shadowed / specular shades
