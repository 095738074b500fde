#!/usr/bin/env python3
"""Per-kernel times of a steady-state rebuild of the bench scene (HIP events on the launch stream), Bool and Vec flavours."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, torch, voxhip, vx_scenes
v, t = vx_scenes.scene("atrium262k")
mesh = voxhip.Mesh.from_arrays(v, t)
vs = np.float32(32.0 / 512)
for kind, name in ((voxhip.GRID_BOOL, "bool"), (voxhip.GRID_VEC, "vec")):
    g = voxhip.Grid.voxelize(mesh, vs, kind)
    for _ in range(3):
        g.revoxelize(mesh, vs)
    torch.cuda.synchronize()
    voxhip.profile_reset(); voxhip.profile_enable(True)
    for _ in range(10):
        g.revoxelize(mesh, vs)
    torch.cuda.synchronize(); voxhip.profile_enable(False)
    k = voxhip.profile_read()
    print(name + ": " + " ".join("%s %.4f" % (nm, ms / c) for nm, (ms, c) in sorted(k.items())))
