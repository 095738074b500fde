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
    if hasattr(L, "vx_debug_walk_hist"): L.vx_debug_walk_hist(None, 1)
    g.trace_device(rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr()); torch.cuda.synchronize()
    L.vx_debug_walk(buf, 1)
    a = list(buf)
    print("rays %d" % n)
    for i, nm in enumerate(names):
        w, l = a[2 * i], a[2 * i + 1]
        print("  %-20s wave-execs %10d (%.2f per ray)  lanes %11d (%.2f per ray)  utilisation %.3f" % (nm, w, w / n, l, l / n, l / (64.0 * w) if w else 0))
    c = a[16:20]; tot = float(sum(c)) or 1.0
    print("  wave cycles: refill %.1f%%  walk %.1f%%  brick %.1f%%  retire %.1f%%   (total %.3g)" % tuple([100 * x / tot for x in c] + [tot]))
    ts = (C.c_ulonglong * (3 * 8192))()
    if hasattr(L, "vx_debug_walk_ts") and L.vx_debug_walk_ts(ts) == 0:
        a = np.frombuffer(ts, dtype=np.uint64).reshape(8192, 3).astype(np.float64)
        a = a[a[:, 2] > 0]
        # (s_memtime bases differ between XCDs: only differences within one wave mean anything)
        d = np.where(a[:, 1] > 0, a[:, 1] - a[:, 0], a[:, 2] - a[:, 0]) / 2400.0
        e = (a[:, 2] - a[:, 0]) / 2400.0
        q = [0, 10, 50, 90, 99, 100]
        print("  waves %d; us (at 2.4 GHz) from a wave's start to: its queue running dry %s | its exit %s" % (len(a), np.percentile(d, q).round(1), np.percentile(e, q).round(1)))
    hb = (C.c_ulonglong * 32)()
    if hasattr(L, "vx_debug_walk_hist") and L.vx_debug_walk_hist(hb, 1) == 0:
        h = list(hb); tot_r = float(sum(h[:16])) or 1.0; tot_s = float(sum(h[16:])) or 1.0
        print("  slab steps per ray (pieces of split rays count separately): " + "  ".join("[%d,%d) %.1f%% of rays, %.1f%% of steps" % (1 << b if b else 0, 1 << (b + 1), 100 * h[b] / tot_r, 100 * h[16 + b] / tot_s) for b in range(16) if h[b]))
