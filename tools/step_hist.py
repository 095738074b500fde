# diagnostic: needs a library built with VOXHIP_EXTRA_FLAGS=-DVX_TRACE_DEBUG_STEPS (t_out then holds step counts)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, voxhip, vx_scenes
v, t = vx_scenes.scene("atrium262k")
g = voxhip.Grid.voxelize(voxhip.Mesh.from_arrays(v, t), np.float32(32 / 512))
rays = vx_scenes.random_rays(1_000_000, v.min(0), v.max(0), seed=2)
steps, _ = g.trace(rays, want_prim=False)
s = np.sort(steps)
print("rays", len(s), "mean", s.mean(), "median", np.median(s), "p90", s[int(.9 * len(s))], "p99", s[int(.99 * len(s))], "p99.9", s[int(.999 * len(s))], "max", s.max())
print("sum steps", s.sum(), "rays>100:", (s > 100).sum(), "rays>300:", (s > 300).sum(), "rays>1000:", (s > 1000).sum())
h, e = np.histogram(s, bins=[0, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096, 1 << 21])
print(list(zip(e[:-1].astype(int), h)))
