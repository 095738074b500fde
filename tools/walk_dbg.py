#!/usr/bin/env python3
"""Reads the lane-utilisation counters of a -DVX_W_DEBUG build of k_walk (tools/variant: VOXHIP_EXTRA_FLAGS=-DVX_W_DEBUG)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, torch, voxhip, vx_scenes
L = voxhip.lib()
buf = (C.c_ulonglong * 24)()
v, t = vx_scenes.scene("atrium262k")
g = voxhip.Grid.voxelize(voxhip.Mesh.from_arrays(v, t), np.float32(32.0 / 512))
names = ["ray set-up", "mip lookup", "slab step", "brick", "cand slab", "exact test", "retire", "round (busy lanes)"]
for n in (1_000_000, 8_000_000):
    rays = torch.from_numpy(vx_scenes.random_rays(n, v.min(0), v.max(0), seed=2)).cuda()
    d_t = torch.empty(n, dtype=torch.float32, device="cuda"); d_p = torch.empty(n, dtype=torch.int32, device="cuda")
    g.trace_device(rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr()); torch.cuda.synchronize()
    L.vx_debug_walk(None, 1)
    g.trace_device(rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr()); torch.cuda.synchronize()
    L.vx_debug_walk(buf, 1)
    a = list(buf)
    print("rays %d" % n)
    for i, nm in enumerate(names):
        w, l = a[2 * i], a[2 * i + 1]
        print("  %-20s wave-execs %10d (%.2f per ray)  lanes %11d (%.2f per ray)  utilisation %.3f" % (nm, w, w / n, l, l / n, l / (64.0 * w) if w else 0))
    c = a[16:20]; tot = float(sum(c)) or 1.0
    print("  wave cycles: refill %.1f%%  walk %.1f%%  brick %.1f%%  retire %.1f%%   (total %.3g)" % tuple([100 * x / tot for x in c] + [tot]))
