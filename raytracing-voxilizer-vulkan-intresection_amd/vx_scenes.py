"""Deterministic synthetic scenes.

The reference ships no geometry (`*.obj` is git-ignored, /root/reference/.gitignore:8; media/scenes holds only
.mtl files), so every mesh named in BASELINE.json's configs is synthesised here from fixed seeds.  All generators
return (verts float32[V,3], tris int32[T,3]); `write_obj` emits `v`/`f i j k` lines with %.9g so that a float32
survives the text round trip exactly.
"""
import numpy as np


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def cube(half=1.0, center=(0.0, 0.0, 0.0)):
    """Axis-aligned cube, 8 vertices / 12 triangles (the cube.obj of BASELINE.json configs[0])."""
    c = np.array(center, dtype=np.float64)
    v = np.array([[x, y, z] for z in (-1, 1) for y in (-1, 1) for x in (-1, 1)], dtype=np.float64) * half + c
    q = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
    t = []
    for a, b, c_, d in q:
        t += [(a, b, c_), (a, c_, d)]
    return _f32(v), np.array(t, dtype=np.int32)


def rotated_cube(half=1.0, angles=(0.37, 0.61, 0.23), offset=(0.113, -0.071, 0.057)):
    """Cube under a fixed non-trivial rotation + offset: no face coincides with a voxel plane."""
    v, t = cube(half)
    ax, ay, az = angles
    rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    r = rz @ ry @ rx
    return _f32(v.astype(np.float64) @ r.T + np.array(offset)), t


def blob(nlon=190, nlat=185, seed=1):
    """Closed lumpy lat-long sphere, 2*nlon*(nlat-1) triangles (default 69 920: the '~70k-tri bunny' stand-in)."""
    rng = np.random.default_rng(seed)
    k = rng.normal(size=(6, 3))
    ph = rng.uniform(0, 2 * np.pi, size=6)
    amp = np.array([0.22, 0.15, 0.11, 0.08, 0.06, 0.04])
    th = np.linspace(0, np.pi, nlat + 1)[1:-1]
    lo = np.linspace(0, 2 * np.pi, nlon, endpoint=False)
    T, L = np.meshgrid(th, lo, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(L), np.cos(T), np.sin(T) * np.sin(L)], axis=-1).reshape(-1, 3)
    poles = np.array([[0, 1.0, 0], [0, -1.0, 0]])
    d = np.concatenate([d, poles])
    r = 1.0 + sum(a * np.sin((d @ kk) * (2.0 + 1.5 * j) + p) for j, (a, kk, p) in enumerate(zip(amp, k, ph)))
    v = d * r[:, None] * np.array([1.0, 1.25, 0.8])
    # normalise so the bbox is exactly [-1,1]^3: voxelsize 2/256 then gives exactly 256^3 cells
    v = (v - v.min(0)) / (v.max(0) - v.min(0)) * 2.0 - 1.0
    nr = nlat - 1
    tris = []
    idx = np.arange(nr * nlon).reshape(nr, nlon)
    a = idx[:-1, :]
    b = np.roll(idx, -1, axis=1)[:-1, :]
    c = idx[1:, :]
    e = np.roll(idx, -1, axis=1)[1:, :]
    tris.append(np.stack([a, c, b], -1).reshape(-1, 3))
    tris.append(np.stack([b, c, e], -1).reshape(-1, 3))
    north, south = nr * nlon, nr * nlon + 1
    tris.append(np.stack([np.full(nlon, north), idx[0], np.roll(idx[0], -1)], -1))
    tris.append(np.stack([np.full(nlon, south), np.roll(idx[-1], -1), idx[-1]], -1))
    return _f32(v), np.concatenate(tris).astype(np.int32)


def _grid_sheet(p00, du, dv, nu, nv, disp=None):
    u = np.linspace(0, 1, nu + 1)
    w = np.linspace(0, 1, nv + 1)
    U, W = np.meshgrid(u, w, indexing="ij")
    p = np.array(p00)[None, None, :] + U[..., None] * np.array(du) + W[..., None] * np.array(dv)
    if disp is not None:
        p = p + disp(U, W)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    t = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return p.reshape(-1, 3), t


def _cylinder(base, radius, height, nseg, nrings, flute=0.0):
    ang = np.linspace(0, 2 * np.pi, nseg, endpoint=False)
    hs = np.linspace(0, height, nrings + 1)
    A, Hh = np.meshgrid(ang, hs, indexing="ij")
    r = radius * (1.0 + flute * np.cos(12 * A))
    p = np.stack([base[0] + r * np.cos(A), base[1] + Hh, base[2] + r * np.sin(A)], -1)
    idx = np.arange(nseg * (nrings + 1)).reshape(nseg, nrings + 1)
    a, b = idx[:, :-1], np.roll(idx, -1, axis=0)[:, :-1]
    c, d = np.roll(idx, -1, axis=0)[:, 1:], idx[:, 1:]
    t = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)])
    return p.reshape(-1, 3), t


def atrium(seed=3, detail=1.0):
    """Sponza-like architectural hall (default ~262k triangles, bbox exactly [-16,16]x[0,32]x[-16,16]): a floor, walls and a ceiling made of a few LARGE
    triangles, two colonnades of fluted columns, arches between them and hanging drapes (finely tessellated
    displaced sheets).  Mixes candidate boxes of 10^1 and 10^7 voxels, which is what stresses the voxelizer's
    load balancing at 512^3 / 1024^3."""
    rng = np.random.default_rng(seed)
    parts = []

    def add(p, t):
        parts.append((np.asarray(p, dtype=np.float64), np.asarray(t, dtype=np.int64)))

    Lx, Ly, Lz = 32.0, 32.0, 32.0   # bbox exactly 32^3: voxelsize 32/512 gives exactly 512^3 cells
    # big flat surfaces: 2 triangles each, deliberately NOT on round coordinates
    add(*_grid_sheet((-Lx / 2, 0.013, -Lz / 2), (Lx, 0, 0), (0, 0, Lz), 1, 1))            # floor
    add(*_grid_sheet((-Lx / 2, Ly - 0.021, -Lz / 2), (Lx, 0, 0), (0, 0, Lz), 1, 1))       # ceiling
    add(*_grid_sheet((-Lx / 2, 0, -Lz / 2 + 0.017), (Lx, 0, 0), (0, Ly, 0), 1, 1))        # back wall
    add(*_grid_sheet((-Lx / 2 + 0.011, 0, -Lz / 2), (0, 0, Lz), (0, Ly, 0), 1, 1))        # left wall
    add(*_grid_sheet((Lx / 2 - 0.019, 0, -Lz / 2), (0, 0, Lz), (0, Ly, 0), 1, 1))         # right wall
    # slanted roof beams: large diagonal triangles
    for k in range(6):
        x0 = -Lx / 2 + (k + 0.5) * Lx / 6
        add(np.array([[x0, Ly * 0.7, -Lz / 2], [x0 + 0.8, Ly * 0.98, 0.1], [x0 - 0.3, Ly * 0.72, Lz / 2]]), np.array([[0, 1, 2]]))
    ncol = 12
    nseg, nrings = int(48 * detail), int(48 * detail)
    for side in (-1, 1):
        for k in range(ncol):
            x = -Lx / 2 + (k + 0.5) * Lx / ncol + rng.uniform(-0.05, 0.05)
            z = side * Lz * 0.27
            add(*_cylinder((x, 0.0, z), 0.62, Ly * 0.55, nseg, nrings, flute=0.04))
            # capital: a squat wider cylinder
            add(*_cylinder((x, Ly * 0.55, z), 0.85, 0.5, nseg, 2))
    # arches between neighbouring columns
    na = int(40 * detail)
    for side in (-1, 1):
        for k in range(ncol - 1):
            x0 = -Lx / 2 + (k + 0.5) * Lx / ncol
            x1 = x0 + Lx / ncol
            z = side * Lz * 0.27

            def arch(U, W, x0=x0, x1=x1):
                th = np.pi * U
                return np.stack([0 * U, 1.1 * np.sin(th), 0 * U], -1)
            add(*_grid_sheet((x0, Ly * 0.55 + 0.5, z - 0.3), (x1 - x0, 0, 0), (0, 0, 0.6), na, 4, arch))
    # drapes: finely tessellated wavy sheets
    nd = int(118 * detail)
    for k in range(5):
        x0 = -Lx / 2 + 2.0 + k * 6.0
        ph = rng.uniform(0, 6.28)

        def wave(U, W, ph=ph):
            return np.stack([0.0 * U, 0.25 * np.sin(5 * U * np.pi + ph) * W, 0.35 * np.sin(9 * U * np.pi + ph) * (0.3 + W)], -1)
        add(*_grid_sheet((x0, Ly * 0.9, -1.0), (3.6, 0, 0), (0, -Ly * 0.5, 0.4), nd, nd, wave))
    vs, ts, off = [], [], 0
    for p, t in parts:
        vs.append(p)
        ts.append(t + off)
        off += p.shape[0]
    return _f32(np.concatenate(vs)), np.concatenate(ts).astype(np.int32)


def soup(ntri, seed=4, edge=0.004, extent=1.0):
    """Random small triangles: centres uniform in [0,extent]^3, vertices within +-edge of the centre.  Vertices are
    not shared (V = 3T).  C5's 10M-triangle soup is soup(10_000_000, 4, edge=1.5/2048)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, extent, size=(ntri, 1, 3))
    v = c + rng.uniform(-edge, edge, size=(ntri, 3, 3))
    v = np.clip(v, 0.0, extent)
    return _f32(v.reshape(-1, 3)), np.arange(3 * ntri, dtype=np.int32).reshape(-1, 3)


def adversarial(seed=7):
    """Knife-edge inputs: walls exactly on multiples of 0.125, zero-area / collinear /
    repeated-vertex triangles, edges <= 1e-9, bbox-spanning diagonal triangles, grid-snapped small triangles."""
    rng = np.random.default_rng(seed)
    V, T = [], []

    def tri(a, b, c):
        n = len(V)
        V.extend([a, b, c])
        T.append((n, n + 1, n + 2))
    for k in range(9):
        x = k * 0.125
        tri((x, 0, 0), (x, 1, 0), (x, 1, 1))
        tri((0, x, 0), (1, x, 0), (1, x, 1))
        tri((0, 0, x), (1, 0, x), (1, 1, x))
    for _ in range(60):
        p = rng.uniform(0, 1, 3)
        q = rng.uniform(0, 1, 3)
        tri(p, p, p)                       # point
        tri(p, q, p)                       # repeated vertex
        tri(p, (p + q) / 2, q)             # collinear
        tri(p, p + 1e-9, p + np.array([0, 1e-9, 0]))  # tiny
    for _ in range(50):
        a = rng.integers(0, 2, 3).astype(float)
        tri(a, 1 - a, rng.uniform(0, 1, 3))  # spans the bbox diagonally
    g = rng.integers(0, 64, size=(2000, 3)) / 64.0
    for p in g:
        d = rng.integers(-2, 3, size=(2, 3)) / 64.0
        tri(p, np.clip(p + d[0], 0, 1), np.clip(p + d[1], 0, 1))
    return _f32(np.array(V)), np.array(T, dtype=np.int32)


def random_rays(n, bmin, bmax, seed=2):
    """origins uniform on a sphere of radius 2x the bbox diagonal around the bbox centre,
    directions toward uniform points inside the bbox (not normalised away from unit length: they ARE normalised).
    No zero direction components (0*inf would give NaN in the slab formula)."""
    rng = np.random.default_rng(seed)
    bmin = np.asarray(bmin, np.float64)
    bmax = np.asarray(bmax, np.float64)
    ctr = (bmin + bmax) / 2
    R = 2.0 * np.linalg.norm(bmax - bmin)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = ctr + R * d
    tgt = rng.uniform(bmin, bmax, size=(n, 3))
    dr = tgt - o
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    dr = dr.astype(np.float32)
    dr[dr == 0] = np.float32(1e-20)
    return np.ascontiguousarray(np.concatenate([o.astype(np.float32), dr], axis=1))


def camera_matrices(eye=(6.16636, 2.42256, -3.15471), ctr=(0.0, 1.0, 0.0), up=(0.0, 1.0, 0.0), fov_deg=60.0,
                    aspect=1280.0 / 720.0, near=0.1, far=1000.0):
    """viewInverse / projInverse as the reference uploads them (hello_vulkan.cpp:69-77; camera main.cpp:92):
    lookAt RH, perspectiveRH_ZO with [1][1] *= -1.  Column-major float32[16] each.  The fov is nvpro_core's
    CameraManip default, third-party and not in the tree (believed 60 deg; unpinned)."""
    eye, ctr, up = (np.array(a, np.float64) for a in (eye, ctr, up))
    f = ctr - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    view = np.eye(4)
    view[0, :3], view[1, :3], view[2, :3] = s, u, -f
    view[:3, 3] = [-s @ eye, -u @ eye, f @ eye]
    t = np.tan(np.radians(fov_deg) / 2)
    proj = np.zeros((4, 4))
    proj[0, 0] = 1 / (aspect * t)
    proj[1, 1] = -1 / t
    proj[2, 2] = far / (near - far)
    proj[3, 2] = -1
    proj[2, 3] = -(far * near) / (far - near)
    vi = np.linalg.inv(view).T.astype(np.float32).reshape(16)   # .T -> column-major
    pi = np.linalg.inv(proj).T.astype(np.float32).reshape(16)
    return vi, pi


# Cameras INSIDE the atrium hall (bbox [-16,16] x [0,32] x [-16,16], open towards +z): what the reference renders -- its camera stands in
# the scene (main.cpp:92) -- as opposed to bench.py's headline batch, whose rays start outside a scene closed on five sides.
INTERIOR_CAMERAS = (dict(eye=(0.0, 6.0, 14.0), ctr=(-2.0, 10.0, -16.0)), dict(eye=(-10.0, 12.0, 10.0), ctr=(8.0, 8.0, -12.0)))


def write_obj(path, verts, tris, header="synthetic scene"):
    v = np.asarray(verts, dtype=np.float32)
    t = np.asarray(tris, dtype=np.int64) + 1
    with open(path, "w") as fh:
        fh.write("# %s\n" % header)
        fh.write("".join("v %.9g %.9g %.9g\n" % (a, b, c) for a, b, c in v.tolist()))
        fh.write("".join("f %d %d %d\n" % (a, b, c) for a, b, c in t.tolist()))


def scene(name):
    """Named scenes used by bench.py / tests / the CLI fixtures."""
    if name == "cube":
        return cube()
    if name == "rotcube":
        return rotated_cube()
    if name == "blob70k":
        return blob()
    if name == "atrium262k":
        return atrium()
    if name == "adversarial":
        return adversarial()
    if name.startswith("soup"):
        n = int(name[4:].replace("k", "000").replace("m", "000000"))
        return soup(n)
    raise KeyError(name)
