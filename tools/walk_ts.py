#!/usr/bin/env python3
"""Per-wave timeline of k_walk from a -DVX_W_TS build (three s_memtime reads per wave, otherwise the product kernel):
when a wave's ray queue ran dry and when the wave left, relative to its own start."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, torch, voxhip, vx_scenes
L = voxhip.lib()
v, t = vx_scenes.scene("atrium262k")
g = voxhip.Grid.voxelize(voxhip.Mesh.from_arrays(v, t), np.float32(32.0 / 512))
ts = (C.c_ulonglong * (3 * 8192))()
for n in [int(x) for x in (sys.argv[1:] or ["1000000", "8000000"])]:
    rays = torch.from_numpy(vx_scenes.random_rays(n, v.min(0), v.max(0), seed=2)).cuda()
    d_t = torch.empty(n, dtype=torch.float32, device="cuda"); d_p = torch.empty(n, dtype=torch.int32, device="cuda")
    for _ in range(3):
        g.trace_device(rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.trace_device(rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr()); e1.record(); torch.cuda.synchronize()
    assert L.vx_debug_walk_ts(ts) == 0
    a = np.frombuffer(ts, dtype=np.uint64).reshape(8192, 3).astype(np.float64)
    a = a[a[:, 2] > 0]
    d = np.where(a[:, 1] > 0, a[:, 1] - a[:, 0], a[:, 2] - a[:, 0]) / 2400.0
    e = (a[:, 2] - a[:, 0]) / 2400.0
    q = [0, 10, 50, 90, 99, 100]
    print("rays %d: trace call %.3f ms; %d waves; us (at 2.4 GHz) from a wave's start to its queue running dry %s | to its exit %s | mean exit %.1f" %
          (n, e0.elapsed_time(e1), len(a), np.percentile(d, q).round(1), np.percentile(e, q).round(1), e.mean()))
