#!/usr/bin/env python3
"""Per-kernel times of a steady-state SHARDED rebuild of the bench scene (word shard 0 of N by rank), HIP events on the launch stream.
   usage: shard_time.py [N=2] [grid=512]      (VOXHIP_VOX_TILED_SHARDS=0 for the direct form)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, torch, voxhip, vx_scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
G = int(sys.argv[2]) if len(sys.argv) > 2 else 512
v, t = vx_scenes.scene("atrium262k")
mesh = voxhip.Mesh.from_arrays(v, t)
vs = np.float32(32.0 / G)
g = voxhip.Grid.voxelize(mesh, vs, voxhip.GRID_BOOL, shard=(0, N))
for _ in range(3):
    g.revoxelize(mesh, vs, shard=(0, N))
torch.cuda.synchronize()
voxhip.profile_reset(); voxhip.profile_enable(True)
for _ in range(10):
    g.revoxelize(mesh, vs, shard=(0, N))
torch.cuda.synchronize(); voxhip.profile_enable(False)
k = voxhip.profile_read()
print("shard 0/%d @ %d^3: " % (N, G) + " ".join("%s %.4f" % (nm, ms / c) for nm, (ms, c) in sorted(k.items())) + " | sum %.4f" % sum(ms / c for ms, c in k.values()))
