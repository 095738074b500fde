#!/usr/bin/env python3
"""Reads the per-phase wave-cycle counters of a -DVX_VOX_DEBUG build (k_voxelize / k_emit_units share for_each_unit)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, torch, voxhip, vx_scenes
L = voxhip.lib()
buf = (C.c_ulonglong * 8)()
v, t = vx_scenes.scene("atrium262k")
mesh = voxhip.Mesh.from_arrays(v, t)
for kind, name in ((voxhip.GRID_BOOL, "bool (k_voxelize only)"), (voxhip.GRID_VEC, "vec (k_voxelize + k_emit_units)")):
    g = voxhip.Grid.voxelize(mesh, np.float32(32.0 / 512), kind)
    g.revoxelize(mesh, np.float32(32.0 / 512)); torch.cuda.synchronize()
    L.vx_debug_vox(None, 1)
    g.revoxelize(mesh, np.float32(32.0 / 512)); torch.cuda.synchronize()
    L.vx_debug_vox(buf, 1)
    a = list(buf)[:5] + [0]; tot = float(sum(a)) or 1.0
    print("%-34s lanes with bits per pass-wave total %d, distinct words among them %d (merge potential %.2fx), atomic requests really sent %d" % (name, buf[6], buf[7], buf[6] / max(1, buf[7]), buf[5]))
    print("%-34s wave cycles: issue next staging %.1f%%  search %.1f%%  functor %.1f%%  rest %.1f%%  wait vmcnt(0) %.1f%%  barrier %.1f%%  (total %.3g)" % ((name,) + tuple(100 * x / tot for x in a) + (tot,)))
