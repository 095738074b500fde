#!/usr/bin/env python3
"""bench.py -- one "step" = one pass of the hot path over one batch of synthetic input, inputs resident in HBM:

    VoxelGridVec-flavoured buildVoxelGrid (bbox, grid dims, SAT voxelization into the occupancy bitmask AND the
    ordered-with-duplicates AABB list)  ->  VoxelGridBool::getAabbs (ascending AABB list)  ->  first-hit trace of R rays.

Workload (N=1): BASELINE.json configs[2] -- the Sponza-like `atrium262k` scene (261 496 triangles, synthetic: the reference
ships no meshes) at voxelsize 32/512 = exactly 512^3 cells, VecEncoding path, with configs[1]'s ray recipe (1M random rays,
tmin 0.001, tmax 1e4).  The north-star target (>=10 Mrays/s on a 512^3 grid) is quoted on this grid size.

N>1 (one process per GPU, torch.distributed / RCCL): every rank voxelizes only its word-aligned shard of the SAME grid,
one all-gather of the shards over xGMI rebuilds the full bitmask on every rank (word-disjoint shards: all-gather == OR),
then every rank traces its own R rays (weak scaling in rays).  value = all ranks' rays / max-over-ranks step time.

Timing: W untimed warm-up steps, an untimed survey pass (HIP events around EVERY kernel launch -> `kernels_survey_pass`, the
dominant kernel), then EXACTLY K timed steps between barrier + synchronize on both sides; inside the timed region only the
dominant kernel is bracketed by events (-> `roofline`), and the stage events are read after the region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch  # first: libvoxhip must bind to the HIP runtime torch ships (see voxhip._preload_torch_hip_runtime)

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="atrium262k")
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--rays", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-ray-sample", type=int, default=1500)
    ap.add_argument("--big-rays", type=int, default=8_000_000, help="extra untimed-for-`value` measurement: trace throughput on a large batch (0 = skip)")
    ap.add_argument("--flavour", default="vec", choices=["vec", "bool"],
                    help="vec: VoxelGridVec build (BASELINE configs[2]); bool: VoxelGridBool build + K4 getAabbs (the app's default path)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; the driver's multi-GPU runs) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses device 0")
    return ap.parse_args()


class DevView:
    """Zero-copy torch view of a raw device pointer (plumbing for the RCCL exchange of the grid's own bitmask)."""

    def __init__(self, ptr, nwords):
        self.__cuda_array_interface__ = {"shape": (nwords,), "typestr": "<i4", "data": (ptr, False), "version": 3, "strides": None}


def main():
    a = parse()
    import voxhip
    import vx_scenes
    import vx_dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    if a.single_device:
        local = 0
    torch.cuda.set_device(local)
    voxhip.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)
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

    kind = voxhip.GRID_VEC if (world == 1 and a.flavour == "vec") else voxhip.GRID_BOOL
    grid = voxhip.Grid.voxelize(mesh, vs, kind)  # sizes the handle's buffers (untimed)
    desc = grid.describe()
    nwords = desc["num_words"]
    wb, we, chunk = voxhip.shard_words(nwords, rank, world)
    # capacity of the getAabbs output: the whole list (VoxelGridVec keeps one Aabb per setVoxel call, duplicates included;
    # the Bool flavour one per occupied voxel -- for N > 1 the unsharded count, the timed loop rebuilds the mask from shards)
    cap = max(desc["set_calls"] if kind == voxhip.GRID_VEC else desc["occupied"], 1)
    d_aabbs = torch.empty(cap * 6, dtype=torch.float32, device=dev)
    gathered = torch.empty(chunk * world, dtype=torch.int32, device=dev) if world > 1 else None

    # stage boundaries: five events per timed step, all read AFTER the timed region (reading them per step needs a device
    # synchronize per step, i.e. ~50 us of idle GPU per 1 ms step that is not part of the workload)
    ev_all = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(a.steps)]
    stage_ms = np.zeros(4)

    def step(timed, k=0):
        ev = ev_all[k] if timed else None
        if timed:
            ev[0].record()
        if world == 1:
            grid.revoxelize(mesh, vs)
        else:
            grid.revoxelize(mesh, vs, words=(wb, we))
        if timed:
            ev[1].record()
        if world > 1:
            mask = torch.as_tensor(DevView(grid.bitmask_device_ptr(mutable=True), nwords), device=dev)
            vx_dist.exchange_bitmask(mask, gathered, wb, we, chunk, dist)
        if timed:
            ev[2].record()
        n = grid.aabbs_device(d_aabbs.data_ptr(), cap)
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
        nocc = step(True, k)
    barrier()
    dt = time.perf_counter() - t0
    for ev in ev_all:
        for k in range(4):
            stage_ms[k] += ev[k].elapsed_time(ev[k + 1])
    voxhip.profile_enable(False)
    kern = voxhip.profile_read()   # the dominant kernel only, measured inside the timed region
    voxhip.profile_select(None)
    stage_ms /= a.steps

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if a.dist_backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt * 1e3 / a.steps
    total_rays = a.rays * world
    value = total_rays * a.steps / dt / 1e6

    hits = int((d_t > 0).sum().item())
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
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (by summed device time over the timed region)
    T, N, R = len(tris), desc["dim"][0] * desc["dim"][1] * desc["dim"][2], a.rays
    gd = grid.describe()
    alg_bytes = {  # algorithmic bytes per launch
        "k_voxelize": 36 * T + 8 * ((N + 31) // 32),
        "k_trace": 28 * R + 4 * ((N + 31) // 32),
        "k_emit_bool": 4 * ((N + 31) // 32) + 24 * gd["occupied"],
        "k_emit_units": 36 * T + 24 * gd["set_calls"],
    }
    roof = None
    if dom is not None:
        ms, n = kern[dom]
        avg_ms = ms / max(n, 1)
        ab = alg_bytes.get(dom)
        ach = (ab / (avg_ms * 1e-3) / 1e9) if ab else None
        traffic = None
        try:  # HBM bytes per launch from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this workload (profiles/, committed)
            with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as fh:
                tj = json.load(fh)
            if tj.get("workload") == a.scene and tj.get("rays") == R and tj.get("grid") == a.grid:
                traffic = tj["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        issue = None
        try:  # instruction-issue view from the SQ_* PMC passes (profiles/, committed): the kernel is not HBM bound
            with open(os.path.join(ROOT, "profiles", "r1_issue.json")) as fh:
                issue = json.load(fh)["kernels"].get(dom)
        except (OSError, ValueError, KeyError):
            pass
        roof = {"bound": "hbm", "kernel": dom, "avg_launch_ms": round(avg_ms, 5), "launches": int(n),
                "algorithmic_bytes": ab, "achieved": round(ach, 2) if ach else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5) if ach else None, "traffic": traffic, "issue": issue,
                "note": "VALU/latency-bound kernel: algorithmic HBM bytes are tiny next to its arithmetic; traffic from rocprofv3 PMC passes is in profiles/"}
    # per-kernel table: the untimed survey pass (every launch bracketed by events)
    kernels = {k: {"avg_ms": round(v[0] / max(v[1], 1), 5), "launches_per_step": round(v[1] / survey_steps, 2)} for k, v in sorted(kern_all.items())}

    cpu = None
    if not a.no_cpu_baseline:
        cpu = cpu_baseline(verts, tris, vs, rays, a.cpu_ray_sample, R)

    out = {
        "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s (%d tris) @ %d^3 grid, %s + %d random rays per GPU"
                   % (a.scene, T, a.grid, "VoxelGridVec build + getAabbs" if kind == voxhip.GRID_VEC else "VoxelGridBool build + getAabbs", R), "grid_dim": list(desc["dim"]), "voxel_size": float(vs), "rays_per_gpu": R,
                   "parallelism": "1 GPU" if world == 1 else "bitmask word-shards x%d + RCCL all-gather, rays independent" % world},
        "mvoxels_per_s": round(N / (stage_ms[0] * 1e-3) / 1e6, 1),
        "mrays_per_s_trace_stage": round(R * world / (stage_ms[3] * 1e-3) / 1e6, 1),
        "stages_ms": {"voxelize": round(float(stage_ms[0]), 4), "exchange": round(float(stage_ms[1]), 4),
                      "get_aabbs": round(float(stage_ms[2]), 4), "trace": round(float(stage_ms[3]), 4)},
        "occupied_voxels": int(nocc), "set_calls": gd["set_calls"], "ray_hits_rank0": hits,
        "trace_large_batch": big,
        "kernel_rooflines": {k: {"achieved_GBps": round(alg_bytes[k] / (kern_all[k][0] / max(kern_all[k][1], 1) * 1e-3) / 1e9, 1),
                                 "frac_of_8TBps": round(alg_bytes[k] / (kern_all[k][0] / max(kern_all[k][1], 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                             for k in alg_bytes if k in kern_all},
        "kernels_survey_pass": kernels, "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(verts, tris, vs, rays, nsample, R):
    """The CPU restatement of the reference path (oracle/, kind "port") timed on this host: serial driver = the
    reference's default (inParaell=false, hello_vulkan.cpp:677), getAabbs, and a brute-force first-hit over a bounded ray
    sample (the reference has no CPU ray path; the brute force is the definition of the result).  Bounded to ~20 s."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    ncores = os.cpu_count() or 1
    t0 = time.perf_counter()
    w, calls, gi = oracle.build_bool(verts, tris, vs)
    t1 = time.perf_counter()
    aabbs = oracle.bool_aabbs(w, gi, vs)
    t2 = time.perf_counter()
    vec = oracle.build_vec(verts, tris, vs, cap=calls)
    t3 = time.perf_counter()
    oracle.trace_brute(aabbs, rays[:nsample], threads=ncores)
    t4 = time.perf_counter()
    build_s, aabb_s, vec_s, trace_s = t1 - t0, t2 - t1, t3 - t2, (t4 - t3)
    step_s = vec_s + aabb_s + trace_s * (R / nsample)
    N = gi["dim"][0] * gi["dim"][1] * gi["dim"][2]
    return {"value": round(R / step_s / 1e6, 6), "unit": "Mrays/s", "cores": ncores, "kind": "port",
            "sample": "full voxelize (serial driver, 1 thread: VoxelGridVec build %.3f s; VoxelGridBool build %.3f s) + getAabbs %.3f s on "
                      "the full %d^3 scene; brute-force first-hit of %d of the %d rays on %d threads (%.3f s), extrapolated linearly"
                      % (vec_s, build_s, aabb_s, gi["dim"][0], nsample, R, ncores, trace_s),
            "voxelize_mvoxels_per_s_1thread": round(N / build_s / 1e6, 2), "brute_force_mrays_per_s": round(nsample / trace_s / 1e6, 6)}


if __name__ == "__main__":
    main()
