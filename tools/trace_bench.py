#!/usr/bin/env python3
"""Ray-stage micro-benchmark on the bench scene (atrium262k @ 512^3): per-kernel times of vx_trace_device at 1M and 8M rays (HIP
events on the launch stream), plus a 300-ray check against the oracle's brute force.  tools/variant_trace.sh runs it per build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, torch, voxhip, vx_scenes
grid = int(os.environ.get("TB_GRID", 512))
v, t = vx_scenes.scene("atrium262k")
vs = np.float32(32.0 / grid)
g = voxhip.Grid.voxelize(voxhip.Mesh.from_arrays(v, t), vs)
out = []
sizes = [int(float(x) * 1_000_000) for x in os.environ.get('TB_SIZES', '1,8').split(',')]
for n in sizes:
    rays_h = vx_scenes.random_rays(n, v.min(0), v.max(0), seed=2)
    rays = torch.from_numpy(rays_h).cuda()
    d_t = torch.empty(n, dtype=torch.float32, device="cuda"); d_p = torch.empty(n, dtype=torch.int32, device="cuda")
    g.trace_device(rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr()); torch.cuda.synchronize()
    voxhip.profile_reset(); voxhip.profile_enable(True)
    reps = 10 if n <= 1_000_000 else 4
    for _ in range(reps):
        g.trace_device(rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr())
    torch.cuda.synchronize(); voxhip.profile_enable(False)
    k = voxhip.profile_read()
    out.append("%gM: " % (n / 1_000_000) + " ".join("%s %.4f" % (nm, ms / c) for nm, (ms, c) in sorted(k.items())))
    if n == 1_000_000 and not os.environ.get("TB_NOCHECK"):
        import oracle
        ow, _, gi = oracle.build_bool(v, t, vs, threads=32)
        oa = oracle.bool_aabbs(ow, gi, vs)
        sel = np.random.default_rng(1).choice(n, 300, replace=False)
        ot, op = oracle.trace_brute(oa, rays_h[sel])
        tt, pp = d_t.cpu().numpy()[sel], d_p.cpu().numpy().view(np.uint32)[sel]
        out.append("check: %s" % ("ok" if (np.array_equal(tt, ot) and np.array_equal(pp, op)) else "MISMATCH %d" % int((tt != ot).sum())))
if os.environ.get("TB_INTERIOR"):
    W, H = 1280, 720
    d_t = torch.empty(W * H, dtype=torch.float32, device="cuda"); d_p = torch.empty(W * H, dtype=torch.int32, device="cuda")
    ms = 0.0
    for cam in vx_scenes.INTERIOR_CAMERAS:
        vi, pi = vx_scenes.camera_matrices(**cam)
        g.trace_primary_device(vi, pi, W, H, d_t.data_ptr(), d_p.data_ptr()); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.trace_primary_device(vi, pi, W, H, d_t.data_ptr(), d_p.data_ptr())
        e1.record(); torch.cuda.synchronize()
        ms += e0.elapsed_time(e1) / 5
    out.append("interior 2x%dx%d: %.4f ms (%.0f Mrays/s)" % (W, H, ms, 2 * W * H / ms / 1e3))
print(" | ".join(out))
