#!/usr/bin/env python3
"""bench.py -- one "step" = one pass of the hot path over one batch of synthetic input, inputs resident in HBM:

    VoxelGridVec-flavoured buildVoxelGrid (bbox, grid dims, SAT voxelization into the occupancy bitmask AND the
    ordered-with-duplicates AABB list)  ->  getAabbs (for the Vec flavour the list the build produced, emitted straight into the
    consumer's device buffer (vx_grid_bind_aabbs_device); `--flavour bool` runs VoxelGridBool::getAabbs = scan + k_emit_bool
    instead)  ->  first-hit trace of R rays.

    The list beside the rays (`list_async` on the line; `--sync-list` switches it off): nothing a step queues after its build reads
    the list -- the ray kernel works on the bitmask's traversal structure -- so the build is asked to leave the kernel that writes the
    list's records to the step's OWN ray batch (VX_VOXELIZE_LIST_ASYNC / vx_grid_aabbs_device_async in include/voxhip.h): getAabbs
    returns the count, the records are written on a low-priority side stream of the grid handle while the ray kernel runs, and the
    next step's build waits for them.  Every step still produces bitmask, list, t and prim before the timed region's closing
    synchronize; the untimed check compares the list the last timed step left in the consumer's buffer byte for byte.

Workload (N=1): BASELINE.json configs[2] -- the Sponza-like `atrium262k` scene (261 496 triangles, synthetic: the reference
ships no meshes) at voxelsize 32/512 = exactly 512^3 cells, VecEncoding path, with configs[1]'s ray recipe (1M random rays,
tmin 0.001, tmax 1e4).  The north-star target (>=10 Mrays/s on a 512^3 grid) is quoted on this grid size.

N>1 (one process per GPU, torch.distributed / RCCL): every rank voxelizes only its word-aligned shard of the SAME grid,
one all-gather of the shards over xGMI rebuilds the full bitmask on every rank (word-disjoint shards: all-gather == OR),
then every rank traces its own R rays (weak scaling in rays).  value = all ranks' rays / max-over-ranks step time.  The
grid stays the N=1 grid so that the per-N values form one curve; BASELINE configs[3] (the same scene at 1024^3, sharded) is
measured beside it, untimed for `value`, as `c4_1024` (`--grid 1024` makes it the timed workload instead).
Started WITHOUT a launcher (`--gpus N`, no WORLD_SIZE in the environment) the script starts its N ranks itself -- child
processes created before this process touches the GPU -- and relays rank 0's line.

Timing: W untimed warm-up steps, an untimed survey pass (HIP events around EVERY kernel launch -> `kernels_survey_pass`, the
dominant kernel), then EXACTLY K timed steps between barrier + synchronize on both sides; inside the timed region only the
dominant kernel is bracketed by events (-> `roofline`), and the stage events are read after the region.  After the timed
region (untimed) a sample of the traced rays is compared with the oracle's brute force, the bitmask with the oracle's and the
AABB list in the consumer's buffer with the oracle's list -> `verified`.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch  # first: libvoxhip must bind to the HIP runtime torch ships (see voxhip._preload_torch_hip_runtime)

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
VALU_ISSUE_PEAK = 1.2288e12  # wave64 VALU instructions/s: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction
PROFILE_TAGS = ("r3", "r2", "r1")  # profiles/<tag>_traffic.json / _issue.json, newest first


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="atrium262k")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--rays", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--cpu-ray-sample", type=int, default=1500)
    ap.add_argument("--cpu-runs", type=int, default=5, help="repetitions of the CPU voxelizer timings (mean and min are reported)")
    ap.add_argument("--verify-rays", type=int, default=1000)
    ap.add_argument("--big-rays", type=int, default=8_000_000, help="extra untimed-for-`value` measurement: trace throughput on a large batch (0 = skip)")
    ap.add_argument("--pipelined-steps", type=int, default=40, help="extra untimed-for-`value` measurement: this many steps with two in flight on two streams (0 = skip)")
    ap.add_argument("--sync-list", action="store_true",
                    help="VoxelGridVec: queue the list's emission inside the build (default: VX_VOXELIZE_LIST_ASYNC -- the build knows the list's length, "
                         "the records are written beside the step's own ray batch, on a low-priority side stream of the grid handle)")
    ap.add_argument("--no-context", action="store_true",
                    help="skip the untimed-for-`value` context block (interior camera on this scene, BASELINE configs[1] and configs[4] trace rates)")
    ap.add_argument("--c4-grid", type=int, default=1024, help="N>1: also measure the sharded build + exchange at this resolution (0 = skip)")
    ap.add_argument("--flavour", default="vec", choices=["vec", "bool"],
                    help="vec: VoxelGridVec build (BASELINE configs[2]); bool: VoxelGridBool build + K4 getAabbs (the app's default path)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; the driver's multi-GPU runs) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses device 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal with one rank: initialise the process group and run the sharded build + exchange path at world size 1 "
                         "(the only way to execute the RCCL calls on a one-GPU box)")
    return ap.parse_args()


def spawn_ranks(a):
    """`--gpus N` without a launcher: start N ranks as fresh child processes (this process has not touched the GPU: no HIP
    call, no torch.cuda.is_available()) and relay rank 0's JSON line.  Never exec: a GPU-initialised process must not be
    replaced, and this one stays around to collect the children."""
    if not a.single_device:
        ndev = torch.cuda.device_count()  # counting devices does not initialise the GPU on this image
        if ndev < a.gpus:
            raise SystemExit("--gpus %d but only %d device(s) visible (use --single-device --dist-backend gloo to rehearse)" % (a.gpus, ndev))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for ln in (out or "").splitlines():
        if ln.startswith("{"):
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if any(rcs) or line is None:
        raise SystemExit("rank exit codes %s, no JSON line from rank 0" % rcs if line is None else "rank exit codes %s" % rcs)
    if json.loads(line).get("n_gpus") != a.gpus:
        raise SystemExit("asked for %d ranks, the job reports %s" % (a.gpus, json.loads(line).get("n_gpus")))
    print(line)


class DevView:
    """Zero-copy torch view of a raw device pointer (plumbing for the RCCL exchange of the grid's own bitmask)."""

    def __init__(self, ptr, nwords):
        self.__cuda_array_interface__ = {"shape": (nwords,), "typestr": "<i4", "data": (ptr, False), "version": 3, "strides": None}


def reference_sat_calls(verts, tris, vs, org, dim):
    """Number of triBoxOverlap calls the reference's loop nest makes (VoxelBuilder.hpp:175-195): the candidate volume of every
    triangle, start = max(0, int((triMin - gridMin) / vs)), end = min(dim, int((triMax - gridMin) / vs) + 2), in float32."""
    p = verts[tris]                                   # (T, 3 vertices, 3 axes)
    tmn, tmx = p.min(axis=1), p.max(axis=1)
    org = org.astype(np.float32)
    s = np.maximum(((tmn - org) / np.float32(vs)).astype(np.int64), 0)
    e = np.minimum(((tmx - org) / np.float32(vs)).astype(np.int64) + 2, np.array(dim, np.int64))
    return int(np.prod(np.maximum(e - s, 0), axis=1).sum())


def load_profile_json(suffix):
    for tag in PROFILE_TAGS:
        try:
            with open(os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, suffix))) as fh:
                return tag, json.load(fh)
        except (OSError, ValueError):
            continue
    return None, None


def trace_context(voxhip, vx_scenes, grid, verts, vs, dev, use_oracle, c5=True):
    """What the headline ray batch does not show (it starts outside a scene closed on five sides: every ray hits the hull).  Untimed
    for `value`, rank 0 at N = 1 only:
      trace_interior  the reference's own use (raytrace.rgen:41-51, main.cpp:92): a camera INSIDE the hall, 1280 x 720, two frames,
                      primary rays generated in the kernel
      trace_c2        BASELINE configs[1]: blob70k at 256^3, 1M random rays
      trace_c5        BASELINE configs[4]: 10M-triangle soup at 2048^3, octree build, 100M coherent primary rays
    each with its hit rate and -- from the oracle's grid walk (the scalar statement of k_walk's traversal) on a sample of the same
    rays -- the mean slab steps per ray."""
    out = {}
    oracle = None
    if use_oracle:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle

    def timed_ms(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def walk_stats(words, gi, vsz, sample):
        if oracle is None or not len(sample):
            return None
        t, p, st = oracle.trace_walk(words, gi, vsz, sample, want_stats=True)
        n = float(len(sample))
        return {"rays_sampled": len(sample), "block_slabs": round(st["block_slabs"] / n, 2), "brick_slabs": round(st["brick_slabs"] / n, 2),
                "cell_slabs": round(st["cell_slabs"] / n, 2), "exact_tests": round(st["exact_tests"] / n, 2),
                "slab_steps": round((st["block_slabs"] + st["brick_slabs"] + st["cell_slabs"]) / n, 2), "hit_rate": round(float((t > 0).mean()), 4)}

    # ---- interior camera on the bench scene
    W, H = 1280, 720
    cams = [vx_scenes.camera_matrices(**c) for c in vx_scenes.INTERIOR_CAMERAS]
    d_t = torch.empty(W * H, dtype=torch.float32, device=dev)
    d_p = torch.empty(W * H, dtype=torch.int32, device=dev)
    ms, hits = 0.0, 0
    for vi, pi in cams:
        ms += timed_ms(lambda: grid.trace_primary_device(vi, pi, W, H, d_t.data_ptr(), d_p.data_ptr()), 5)
        hits += int((d_t > 0).sum().item())
    n = len(cams) * W * H
    st = None
    if oracle is not None:
        gi = oracle.grid_info(verts, vs)
        pix = np.random.default_rng(5).choice(W * H, 20000, replace=False).astype(np.uint64)
        st = walk_stats(grid.bitmask(), gi, vs, np.concatenate([oracle.primary_rays_pixels(vi, pi, W, H, pix) for vi, pi in cams]))
    out["trace_interior"] = {"workload": "camera inside the hall (raytrace.rgen camera model), %d frames of %dx%d primary rays generated in the kernel" % (len(cams), W, H),
                             "rays": n, "ms": round(ms, 4), "mrays_per_s": round(n / (ms * 1e-3) / 1e6, 1), "hit_rate": round(hits / n, 4), "walk": st}
    del d_t, d_p
    # ---- BASELINE configs[1]
    v2, t2 = vx_scenes.scene("blob70k")
    vs2 = np.float32(2.0 / 256)
    g2 = voxhip.Grid.voxelize(voxhip.Mesh.from_arrays(v2, t2), vs2)
    r2 = vx_scenes.random_rays(1_000_000, v2.min(0), v2.max(0), seed=2)
    d_r2 = torch.from_numpy(r2).to(dev)
    d_t2 = torch.empty(len(r2), dtype=torch.float32, device=dev)
    d_p2 = torch.empty(len(r2), dtype=torch.int32, device=dev)
    ms2 = timed_ms(lambda: g2.trace_device(d_r2.data_ptr(), len(r2), d_t2.data_ptr(), d_p2.data_ptr()), 5)
    st2 = walk_stats(g2.bitmask(), oracle.grid_info(v2, vs2), vs2, r2[:20000]) if oracle is not None else None
    out["trace_c2"] = {"workload": "BASELINE configs[1]: blob70k (%d tris) @ 256^3, 1M random rays" % len(t2), "grid_dim": list(g2.describe()["dim"]), "rays": len(r2),
                       "ms": round(ms2, 4), "mrays_per_s": round(len(r2) / (ms2 * 1e-3) / 1e6, 1), "hit_rate": round(float((d_t2 > 0).float().mean().item()), 4), "walk": st2}
    del g2, d_r2, d_t2, d_p2
    # ---- BASELINE configs[4]
    if c5:
        NT, G = 10_000_000, 2048
        v5, t5 = vx_scenes.soup(NT, seed=4, edge=1.5 / G)
        vs5 = np.float32(1.0 / G)
        dv, dt_ = torch.from_numpy(v5).to(dev), torch.from_numpy(t5).to(dev)
        m5 = voxhip.Mesh.from_device(dv.data_ptr(), len(v5), dt_.data_ptr(), len(t5), keep=(dv, dt_))
        torch.cuda.synchronize()
        g5 = voxhip.Grid.voxelize(m5, vs5, voxhip.GRID_BOOL)
        t0 = time.perf_counter()
        g5.revoxelize(m5, vs5)
        d5 = g5.describe()
        build_ms = (time.perf_counter() - t0) * 1e3
        o = voxhip.Octree(m5, vs5)
        del o
        tb = []
        for _ in range(3):
            t0 = time.perf_counter()
            o = voxhip.Octree(m5, vs5)
            tb.append((time.perf_counter() - t0) * 1e3)
            ni, nn = o.num_items, o.num_nodes
            del o
        vi, pi = vx_scenes.camera_matrices(eye=(1.55, 1.25, -0.85), ctr=(0.5, 0.5, 0.5), fov_deg=38.0, aspect=1.0)  # (tests/test_gpu_configs.py: C5)
        W5, H5, tiles = 10000, 10000, 1
        d_t5 = torch.empty(W5 * H5, dtype=torch.float32, device=dev)
        d_p5 = torch.empty(W5 * H5, dtype=torch.int32, device=dev)
        ms5 = timed_ms(lambda: g5.trace_primary_device(vi, pi, W5, H5, d_t5.data_ptr(), d_p5.data_ptr()), 2)
        hr5 = float((d_t5 > 0).float().mean().item())
        st5 = None
        if oracle is not None:
            pix = np.random.default_rng(6).choice(W5 * H5, 20000, replace=False).astype(np.uint64)
            st5 = walk_stats(g5.bitmask(), oracle.grid_info(v5, vs5), vs5, oracle.primary_rays_pixels(vi, pi, W5, H5, pix))
        nv = float(np.prod(d5["dim"]))
        out["trace_c5"] = {"workload": "BASELINE configs[4]: 10M-triangle soup @ 2048^3, octree (sparse) path, %d launch of %dx%d coherent primary rays" % (tiles, W5, H5),
                           "grid_dim": list(d5["dim"]), "occupied_voxels": d5["occupied"], "voxelize_bool_ms_host_clock": round(build_ms, 3),
                           "mvoxels_per_s": round(nv / (build_ms * 1e-3) / 1e6, 1), "octree_build_ms_host_clock": [round(x, 3) for x in tb], "octree_items": ni,
                           "octree_nodes": nn, "rays": tiles * W5 * H5, "ms": round(ms5, 3), "mrays_per_s": round(tiles * W5 * H5 / (ms5 * 1e-3) / 1e6, 1),
                           "hit_rate": round(hr5, 4), "walk": st5}
        del g5, m5, dv, dt_, d_t5, d_p5
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a)
    import voxhip
    import vx_scenes
    import vx_dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if a.single_device:
        local = 0
    torch.cuda.set_device(local)
    voxhip.set_device(local)
    dist = None
    sharded = world > 1 or a.force_dist
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)
        if dist.get_world_size() != a.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), a.gpus))
    dev = torch.device("cuda", local)

    # ---- synthetic inputs, resident in HBM before anything is timed
    verts, tris = vx_scenes.scene(a.scene)
    ext = float((verts.max(0) - verts.min(0)).max())
    vs = np.float32(ext / a.grid)
    d_verts = torch.from_numpy(verts).to(dev)
    d_tris = torch.from_numpy(tris).to(dev)
    mesh = voxhip.Mesh.from_device(d_verts.data_ptr(), len(verts), d_tris.data_ptr(), len(tris), keep=(d_verts, d_tris))
    rays = vx_scenes.random_rays(a.rays, verts.min(0), verts.max(0), seed=2 + rank)
    d_rays = torch.from_numpy(rays).to(dev)
    d_t = torch.empty(a.rays, dtype=torch.float32, device=dev)
    d_prim = torch.empty(a.rays, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    kind = voxhip.GRID_VEC if (not sharded and a.flavour == "vec") else voxhip.GRID_BOOL
    grid = voxhip.Grid.voxelize(mesh, vs, kind)  # sizes the handle's buffers (untimed)
    desc = grid.describe()
    nwords = desc["num_words"]
    wb, we, chunk = voxhip.shard_words(nwords, rank, world)
    # capacity of the getAabbs output: the whole list (VoxelGridVec keeps one Aabb per setVoxel call, duplicates included;
    # the Bool flavour one per occupied voxel -- for N > 1 the unsharded count, the timed loop rebuilds the mask from shards)
    cap = max(desc["set_calls"] if kind == voxhip.GRID_VEC else desc["occupied"], 1)
    d_aabbs = torch.empty(cap * 6, dtype=torch.float32, device=dev)
    if kind == voxhip.GRID_VEC:
        grid.bind_aabbs_device(d_aabbs.data_ptr(), cap)   # the list is built in the consumer's buffer: getAabbs has nothing left to copy
    exch = vx_dist.Exchange(nwords, rank, world, dev, dist) if sharded else None

    # stage boundaries: five events per step of a separate pass AFTER the timed region (the timed steps carry none), all read at
    # its end (reading them per step needs a device synchronize per step, i.e. ~50 us of idle GPU per step)
    stage_steps = min(10, a.steps)
    ev_all = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(stage_steps)]
    stage_ms = np.zeros(4)

    list_async = kind == voxhip.GRID_VEC and not sharded and not a.sync_list
    # VoxelGridBool::getAabbs (the N > 1 step and --flavour bool): the same deferral through vx_grid_aabbs_device_async
    bool_async = kind != voxhip.GRID_VEC and not a.sync_list

    def step(staged, k=0):
        timed = staged
        ev = ev_all[k] if timed else None
        if timed:
            ev[0].record()
        if not sharded:
            grid.revoxelize(mesh, vs, list_async=list_async)
        else:
            grid.revoxelize(mesh, vs, words=(wb, we))
        if timed:
            ev[1].record()
        if sharded:
            mask = torch.as_tensor(DevView(grid.bitmask_device_ptr(mutable=True), nwords), device=dev)
            exch.run(mask)
        if timed:
            ev[2].record()
        n = (grid.aabbs_device_async if bool_async else grid.aabbs_device)(d_aabbs.data_ptr(), cap)
        if timed:
            ev[3].record()
        grid.trace_device(d_rays.data_ptr(), a.rays, d_t.data_ptr(), d_prim.data_ptr())
        if timed:
            ev[4].record()
        return n

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    barrier()
    # survey pass (untimed): HIP events around EVERY kernel launch -> per-kernel table and the dominant kernel.  Two event
    # records per launch cost a few microseconds each, ~45 launches per step, so the timed region below only brackets the
    # dominant kernel (the one the roofline object prices).
    voxhip.profile_reset()
    voxhip.profile_select(None)
    voxhip.profile_enable(True)
    survey_steps = 3
    for _ in range(survey_steps):
        step(False)
    torch.cuda.synchronize()
    voxhip.profile_enable(False)
    kern_all = voxhip.profile_read()
    dom = max(kern_all.items(), key=lambda kv: kv[1][0])[0] if kern_all else None
    barrier()
    voxhip.profile_reset()
    voxhip.profile_select(dom)
    voxhip.profile_enable(True)   # HIP events on the launch stream around the dominant kernel, over the timed region
    t0 = time.perf_counter()
    nocc = 0
    for k in range(a.steps):
        nocc = step(False)   # the timed steps carry no stage events (five event records per step are ~1 % of a step)
    barrier()
    dt = time.perf_counter() - t0
    voxhip.profile_enable(False)
    kern = voxhip.profile_read()   # the dominant kernel only, measured inside the timed region
    voxhip.profile_select(None)
    # stage breakdown: a pass of its own after the timed region, five events per step
    for k in range(stage_steps):
        step(True, k)
    barrier()
    for ev in ev_all:
        for k in range(4):
            stage_ms[k] += ev[k].elapsed_time(ev[k + 1])
    stage_ms /= stage_steps

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if a.dist_backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt * 1e3 / a.steps
    total_rays = a.rays * world
    value = total_rays * a.steps / dt / 1e6

    hits = int((d_t > 0).sum().item())
    h_t = d_t.cpu().numpy()
    h_prim = d_prim.cpu().numpy().view(np.uint32)

    # ---- BASELINE configs[3] beside the timed workload (N > 1): sharded build + exchange of the same scene at 1024^3
    c4 = None
    if sharded and a.c4_grid and a.c4_grid != a.grid:
        vs4 = np.float32(ext / a.c4_grid)
        g4 = voxhip.Grid.voxelize(mesh, vs4, voxhip.GRID_BOOL)
        d4 = g4.describe()
        nw4 = d4["num_words"]
        b4, e4, _ = voxhip.shard_words(nw4, rank, world)
        ex4 = vx_dist.Exchange(nw4, rank, world, dev, dist)
        ev4 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        acc = np.zeros(2)
        reps = 5
        for k in range(reps + 1):
            barrier()
            ev4[0].record()
            g4.revoxelize(mesh, vs4, words=(b4, e4))
            ev4[1].record()
            ex4.run(torch.as_tensor(DevView(g4.bitmask_device_ptr(mutable=True), nw4), device=dev))
            ev4[2].record()
            torch.cuda.synchronize()
            if k:
                acc += (ev4[0].elapsed_time(ev4[1]), ev4[1].elapsed_time(ev4[2]))
        acc /= reps
        tt = torch.tensor(acc, dtype=torch.float64, device=dev if a.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        acc = tt.cpu().numpy()
        g4.refresh()
        occ4 = g4.describe()["occupied"]
        n4 = int(np.prod(d4["dim"]))
        c4 = {"workload": "%s @ %d^3, %d word shards + all-gather" % (a.scene, a.c4_grid, world), "grid_dim": list(d4["dim"]),
              "voxelize_ms": round(float(acc[0]), 4), "exchange_ms": round(float(acc[1]), 4), "exchange_bytes_per_rank": int(ex4.bytes_per_rank),
              "exchange_algo": ex4.algo, "mvoxels_per_s": round(n4 / ((acc[0] + acc[1]) * 1e-3) / 1e6, 1), "occupied_voxels": int(occ4)}
        del g4

    # throughput regime of the ray kernel: at 1M rays the persistent kernel holds only 4 rays per lane and is dominated by its
    # ramp/tail; a large batch shows the steady-state rate (reported separately, never as `value`)
    big = None
    if a.big_rays and rank == 0:
        rb = vx_scenes.random_rays(a.big_rays, verts.min(0), verts.max(0), seed=77)
        d_rb = torch.from_numpy(rb).to(dev)
        d_tb = torch.empty(a.big_rays, dtype=torch.float32, device=dev)
        d_pb = torch.empty(a.big_rays, dtype=torch.int32, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        grid.trace_device(d_rb.data_ptr(), a.big_rays, d_tb.data_ptr(), d_pb.data_ptr())
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            grid.trace_device(d_rb.data_ptr(), a.big_rays, d_tb.data_ptr(), d_pb.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        big = {"rays": a.big_rays, "ms": round(e0.elapsed_time(e1) / 3, 4), "mrays_per_s": round(a.big_rays * 3 / (e0.elapsed_time(e1) * 1e-3) / 1e6, 1)}
        del d_rb, d_tb, d_pb
    # Throughput with TWO steps in flight (untimed for `value`; rank 0, N = 1): two grid handles on two HIP streams, each driven by its
    # own host thread, every step complete and independent (its own build, list, trace and outputs).  The small latency-bound kernels at
    # the head of a build and the drain of a ray launch leave most of the machine idle; a second step fills it.
    piped = None
    if rank == 0 and world == 1 and not sharded and a.pipelined_steps > 0:
        import threading
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        bufs = []
        for k in range(2):
            gk = voxhip.Grid.voxelize(mesh, vs, kind, stream=streams[k].cuda_stream)
            ak = torch.empty(cap * 6, dtype=torch.float32, device=dev)
            if kind == voxhip.GRID_VEC:
                gk.bind_aabbs_device(ak.data_ptr(), cap)
            bufs.append((gk, ak, torch.empty(a.rays, dtype=torch.float32, device=dev), torch.empty(a.rays, dtype=torch.int32, device=dev)))
        torch.cuda.synchronize()

        def worker(k, nsteps):
            torch.cuda.set_device(local)
            voxhip.set_device(local)
            gk, ak, tk, pk = bufs[k]
            for _ in range(nsteps):
                gk.revoxelize(mesh, vs, stream=streams[k].cuda_stream)
                gk.aabbs_device(ak.data_ptr(), cap)
                gk.trace_device(d_rays.data_ptr(), a.rays, tk.data_ptr(), pk.data_ptr())

        def run(nsteps):
            th = [threading.Thread(target=worker, args=(k, nsteps)) for k in range(2)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            torch.cuda.synchronize()
        run(3)
        t0 = time.perf_counter()
        run(a.pipelined_steps // 2)
        dtp = time.perf_counter() - t0
        nst = 2 * (a.pipelined_steps // 2)
        same = bool(torch.equal(bufs[0][2], d_t) and torch.equal(bufs[1][2], d_t) and torch.equal(bufs[0][3], d_prim) and torch.equal(bufs[1][3], d_prim))
        piped = {"what": "the same step, two in flight: two grid handles on two HIP streams, one host thread each (every step builds, lists and traces its own grid)",
                 "steps": nst, "ms_per_step": round(dtp * 1e3 / nst, 4), "mrays_per_s": round(a.rays * nst / dtp / 1e6, 1), "outputs_equal_to_the_timed_steps": same}
        for gk, _, _, _ in bufs:
            gk.free()
        del bufs
    context = None
    if rank == 0 and world == 1 and not sharded and not a.no_context and a.scene == "atrium262k":
        context = trace_context(voxhip, vx_scenes, grid, verts, vs, dev, use_oracle=not a.no_cpu_baseline)
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (by summed device time over the timed region)
    T, N, R = len(tris), desc["dim"][0] * desc["dim"][1] * desc["dim"][2], a.rays
    gd = grid.describe()
    alg_bytes = {  # algorithmic bytes per launch
        "k_voxelize": 36 * T + 8 * ((N + 31) // 32),
        "k_walk": 28 * R + 4 * ((N + 31) // 32),
        "k_emit_bool": 4 * ((N + 31) // 32) + 24 * gd["occupied"],
        "k_emit_units": 36 * T + 24 * gd["set_calls"],
    }
    roof = roof_issue = None
    if dom is not None:
        ms, n = kern[dom]
        avg_ms = ms / max(n, 1)
        ab = alg_bytes.get(dom)
        ach = (ab / (avg_ms * 1e-3) / 1e9) if ab else None
        same_workload = a.scene == "atrium262k" and R == 1_000_000 and a.grid == 512 and world == 1
        traffic = None
        ttag, tj = load_profile_json("traffic")
        if tj and same_workload:  # HBM bytes per launch from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this workload (profiles/, committed)
            traffic = tj["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
        itag, ij = load_profile_json("issue")
        issue = ij["kernels"].get(dom) if (ij and same_workload) else None
        limiter = "hbm"
        if issue:
            # what the PMC passes say bounds the kernel: instruction issue / latency when the algorithmic-bytes fraction is tiny
            wps = issue["valu_wave_insts"] / (avg_ms * 1e-3)
            roof_issue = {"bound": "valu_issue", "kernel": dom, "achieved": round(wps, 1), "peak": VALU_ISSUE_PEAK, "unit": "wave64 VALU instr/s",
                          "frac": round(wps / VALU_ISSUE_PEAK, 4), "lane_utilisation": issue.get("lane_utilisation"),
                          "wait_share_of_wave_cycles": issue.get("wait_share_of_wave_cycles"), "valu_wave_insts_per_launch": issue["valu_wave_insts"],
                          "source": "profiles/%s_issue.json (rocprofv3 SQ_* passes of this command) / live avg launch time" % itag}
            if ach is not None and ach / HBM_PEAK_GBS < 0.05:
                limiter = "valu_issue+latency"
        roof = {"bound": "hbm", "limiter": limiter, "kernel": dom, "avg_launch_ms": round(avg_ms, 5), "launches": int(n),
                "algorithmic_bytes": ab, "achieved": round(ach, 2) if ach else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5) if ach else None, "traffic": traffic,
                "note": "achieved = SURVEY 8(d) algorithmic bytes / live launch time; `limiter` is what the PMC counters say bounds the kernel "
                        "(see roofline_issue); traffic from the rocprofv3 FETCH_SIZE/WRITE_SIZE passes in profiles/%s_traffic.json" % ttag}
    # per-kernel table: the untimed survey pass (every launch bracketed by events)
    kernels = {k: {"avg_ms": round(v[0] / max(v[1], 1), 5), "launches_per_step": round(v[1] / survey_steps, 2)} for k, v in sorted(kern_all.items())}

    cpu, verified = None, None
    oa = None
    if not a.no_cpu_baseline or not a.no_verify:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        ncores = os.cpu_count() or 1
        ow, calls, gi = oracle.build_bool(verts, tris, vs, threads=min(ncores, 32))
        oa = oracle.bool_aabbs(ow, gi, vs)
        if not a.no_verify:
            # untimed: a sample of the rays the timed steps traced, against the brute force over all occupied boxes
            sel = np.random.default_rng(123).choice(R, min(a.verify_rays, R), replace=False)
            ot, op = oracle.trace_brute(oa, rays[sel], threads=ncores)
            words = grid.bitmask()
            verified = bool(np.array_equal(h_t[sel], ot) and np.array_equal(h_prim[sel], op) and np.array_equal(words, ow) and int(nocc) == (gd["set_calls"] if kind == voxhip.GRID_VEC else len(oa)))
            if kind != voxhip.GRID_VEC:
                # VoxelGridBool::getAabbs as the last timed step left it in the consumer's buffer, byte for byte
                torch.cuda.synchronize()
                got = d_aabbs[: int(nocc) * 6].cpu().numpy()
                verified = verified and len(oa) == int(nocc) and got.tobytes() == oa.tobytes()
            if kind == voxhip.GRID_VEC and not sharded:
                # the list itself, as the last timed step left it in the consumer's buffer (with --sync-list off: written beside that step's
                # ray batch), byte for byte against the oracle's VoxelGridVec list
                ov = oracle.build_vec(verts, tris, vs, threads=min(ncores, 32), cap=int(nocc))
                torch.cuda.synchronize()
                got = d_aabbs[: int(nocc) * 6].cpu().numpy()
                verified = verified and len(ov) == int(nocc) and got.tobytes() == ov.tobytes()
        if not a.no_cpu_baseline:
            cpu = cpu_baseline(oracle, verts, tris, vs, rays, oa, a.cpu_ray_sample, R, a.cpu_runs)

    vox_s = stage_ms[0] * 1e-3
    sat_calls = reference_sat_calls(verts, tris, vs, desc["bbox_min"], desc["dim"])
    out = {
        "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "verified": verified, "list_async": bool(list_async or bool_async),
        "config": {"workload": "%s (%d tris) @ %d^3 grid, %s + %d random rays per GPU"
                   % (a.scene, T, a.grid, "VoxelGridVec build + getAabbs" if kind == voxhip.GRID_VEC else "VoxelGridBool build + getAabbs", R), "grid_dim": list(desc["dim"]), "voxel_size": float(vs), "rays_per_gpu": R,
                   "parallelism": "1 GPU" if world == 1 else "bitmask word-shards x%d + RCCL all-gather, rays independent" % world},
        "mvoxels_per_s": round(N / vox_s / 1e6, 1),
        "occupied_voxels_per_s": round(gd["occupied"] / vox_s, 1), "set_voxel_calls_per_s": round(gd["set_calls"] / vox_s, 1),
        "sat_tests_per_s": round(sat_calls / vox_s, 1), "sat_tests_reference_loop": sat_calls,
        "mrays_per_s_trace_stage": round(R * world / (stage_ms[3] * 1e-3) / 1e6, 1),
        "stages_ms": {"voxelize": round(float(stage_ms[0]), 4), "exchange": round(float(stage_ms[1]), 4),
                      "get_aabbs": round(float(stage_ms[2]), 4), "trace": round(float(stage_ms[3]), 4)},
        "occupied_voxels": gd["occupied"], "aabbs_returned": int(nocc), "set_calls": gd["set_calls"], "ray_hits_rank0": hits,
        "trace_large_batch": big, "two_steps_in_flight": piped, "trace_context": context,
        "kernel_rooflines": {k: {"achieved_GBps": round(alg_bytes[k] / (kern_all[k][0] / max(kern_all[k][1], 1) * 1e-3) / 1e9, 1),
                                 "frac_of_8TBps": round(alg_bytes[k] / (kern_all[k][0] / max(kern_all[k][1], 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                             for k in alg_bytes if k in kern_all},
        "kernels_survey_pass": kernels, "roofline": roof, "roofline_issue": roof_issue, "cpu_baseline": cpu,
    }
    # k_voxelize is bound by neither HBM bytes nor issue: its atomicOr lanes are executed at the memory side, one 64-byte request per
    # lane (MI355X_MICROARCH.md, global atomics: ~1.3 TB/s of 64-byte requests chip-wide = ~20.3 G requests/s).  The request count of
    # this workload comes from the -DVX_VOX_DEBUG counters (tools/vox_dbg.py), committed in profiles/.
    atag, aj = load_profile_json("atomics")
    if aj and a.scene == "atrium262k" and a.grid == 512 and world == 1 and "k_voxelize" in kern_all:
        avg = kern_all["k_voxelize"][0] / max(kern_all["k_voxelize"][1], 1) * 1e-3
        req = aj["k_voxelize"]["atomic_lane_requests_per_launch"]
        kv = {"requests_per_launch": req, "achieved_Greq_per_s": round(req / avg / 1e9, 2), "peak_Greq_per_s": aj["peak_Greq_per_s"],
              "frac_of_request_rate": round(req / avg / 1e9 / aj["peak_Greq_per_s"], 3), "source": "profiles/%s_atomics.json" % atag}
        # since the tiled build mask (round 3) the requests are few enough: the kernel is bound by VALU issue (the SAT sweep)
        vtag, vj = load_profile_json("issue")
        vi = ((vj or {}).get("kernels") or {}).get("k_voxelize") if not sharded else None
        if vi:
            wps = vi["valu_wave_insts"] / avg
            kv.update({"bound": "valu_issue", "valu_wave_insts_per_launch": vi["valu_wave_insts"], "frac": round(wps / VALU_ISSUE_PEAK, 4),
                       "lane_utilisation": vi.get("lane_utilisation"), "issue_source": "profiles/%s_issue.json" % vtag})
        else:
            kv.update({"bound": "memory-side atomic requests", "frac": kv["frac_of_request_rate"]})
        out["kernel_rooflines"]["k_voxelize"].update(kv)
    if sharded:
        out["rccl_world"] = dist.get_world_size()
        out["exchange"] = {"algo": exch.algo, "bytes_per_rank": int(exch.bytes_per_rank), "ms": round(float(stage_ms[1]), 4),
                           "GBps_per_rank": round(exch.bytes_per_rank * (world - 1) / max(stage_ms[1], 1e-9) / 1e6, 2)}
        out["c4_1024"] = c4
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(oracle, verts, tris, vs, rays, oa, nsample, R, runs):
    """The CPU restatement of the reference path (oracle/, kind "port") timed on this host with the Benchmaker method
    (hello_vulkan.h:181-240: N runs, mean; min beside it): the serial driver = the reference's default (inParaell=false,
    hello_vulkan.cpp:677), the threaded driver (VoxelBuilder.hpp:424-541) on every core, getAabbs -- and, because the
    reference has NO CPU ray path (its rays run in the Vulkan RT pipeline), two labelled stand-ins for the ray stage: the
    brute force that defines the result, and the oracle's grid-walking tracer when it is built."""
    ncores = os.cpu_count() or 1

    def timed(fn, n):
        ts = []
        res = None
        for _ in range(n):
            t0 = time.perf_counter()
            res = fn()
            ts.append(time.perf_counter() - t0)
        return res, float(np.mean(ts)), float(np.min(ts))

    (w, calls, gi), ser_mean, ser_min = timed(lambda: oracle.build_bool(verts, tris, vs), runs)
    _, thr_mean, thr_min = timed(lambda: oracle.build_bool(verts, tris, vs, threads=ncores), runs)
    _, vec_mean, vec_min = timed(lambda: oracle.build_vec(verts, tris, vs, cap=calls), max(2, runs // 2))
    _, aabb_mean, aabb_min = timed(lambda: oracle.bool_aabbs(w, gi, vs), runs)
    t0 = time.perf_counter()
    oracle.trace_brute(oa, rays[:nsample], threads=ncores)
    brute_s = time.perf_counter() - t0
    N = gi["dim"][0] * gi["dim"][1] * gi["dim"][2]
    walk = None
    if hasattr(oracle, "trace_walk"):
        nw = min(R, 200_000)
        t0 = time.perf_counter()
        oracle.trace_walk(w, gi, vs, rays[:nw], threads=ncores)
        walk_s = time.perf_counter() - t0
        walk = {"rays": nw, "seconds": round(walk_s, 4), "mrays_per_s": round(nw / walk_s / 1e6, 4), "threads": ncores}
    ray_s = (walk["seconds"] * R / walk["rays"]) if walk else brute_s * (R / nsample)
    step_s = vec_mean + aabb_mean + ray_s
    return {"value": round(R / step_s / 1e6, 6), "unit": "Mrays/s", "cores": ncores, "kind": "port", "runs": runs,
            "sample": "voxelizer on the full %d^3 scene, %d runs each: serial driver (1 thread) VoxelGridBool %.3f s mean / %.3f s min, "
                      "VoxelGridVec %.3f / %.3f s; threaded driver (%d threads) %.3f / %.3f s; getAabbs %.4f / %.4f s.  Rays (the "
                      "reference has no CPU ray path): %s; `value` = rays / (VoxelGridVec build + getAabbs + ray stage)"
                      % (gi["dim"][0], runs, ser_mean, ser_min, vec_mean, vec_min, ncores, thr_mean, thr_min, aabb_mean, aabb_min,
                         ("grid-walking CPU tracer on %d of the %d rays, %d threads, %.3f s, scaled linearly" % (walk["rays"], R, ncores, walk["seconds"]))
                         if walk else ("brute force over all boxes on %d of the %d rays, %d threads, %.3f s, scaled linearly" % (nsample, R, ncores, brute_s))),
            "voxelize_serial_s": {"mean": round(ser_mean, 4), "min": round(ser_min, 4)},
            "voxelize_threaded_s": {"mean": round(thr_mean, 4), "min": round(thr_min, 4), "threads": ncores},
            "voxelize_vec_serial_s": {"mean": round(vec_mean, 4), "min": round(vec_min, 4)},
            "get_aabbs_s": {"mean": round(aabb_mean, 5), "min": round(aabb_min, 5)},
            "voxelize_mvoxels_per_s_1thread": round(N / ser_mean / 1e6, 2), "voxelize_mvoxels_per_s_threaded": round(N / thr_mean / 1e6, 2),
            "brute_force_mrays_per_s": round(nsample / brute_s / 1e6, 6), "cpu_ray_walk": walk}


if __name__ == "__main__":
    main()
