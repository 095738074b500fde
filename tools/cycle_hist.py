# diagnostic: needs a library built with VOXHIP_EXTRA_FLAGS=-DVX_TRACE_DEBUG_CYCLES (t_out = per-ray residency, 100 MHz ticks)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, voxhip, vx_scenes
v, t = vx_scenes.scene("atrium262k")
g = voxhip.Grid.voxelize(voxhip.Mesh.from_arrays(v, t), np.float32(32 / 512))
rays = vx_scenes.random_rays(1_000_000, v.min(0), v.max(0), seed=2)
g.trace(rays, want_prim=False)
c, _ = g.trace(rays, want_prim=False)
us = c / 100.0
s = np.sort(us)
print("per-ray residency us: mean %.2f median %.2f p99 %.2f p99.9 %.2f max %.2f" % (s.mean(), np.median(s), s[int(.99 * len(s))], s[int(.999 * len(s))], s.max()))
top = np.argsort(us)[-8:]
for i in top:
    print(i, us[i], rays[i])
