# how much does the compacted hit list (vx_hit records + count) cost on top of t / prim?  (k_rank's compaction path)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, torch, voxhip, vx_scenes
n = 1_000_000
v, t = vx_scenes.scene("atrium262k")
g = voxhip.Grid.voxelize(voxhip.Mesh.from_arrays(v, t), np.float32(32 / 512))
rays = torch.from_numpy(vx_scenes.random_rays(n, v.min(0), v.max(0), seed=2)).cuda()
d_t = torch.empty(n, dtype=torch.float32, device="cuda"); d_p = torch.empty(n, dtype=torch.int32, device="cuda")
d_h = torch.empty(n * 3, dtype=torch.int32, device="cuda"); d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
for hits in (False, True):
    args = (rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr()) + ((d_h.data_ptr(), d_n.data_ptr()) if hits else ())
    g.trace_device(*args); torch.cuda.synchronize()
    voxhip.profile_reset(); voxhip.profile_enable(True)
    for _ in range(5):
        g.trace_device(*args)
    torch.cuda.synchronize(); voxhip.profile_enable(False)
    k = voxhip.profile_read()
    print("hit list" if hits else "t + prim ", {n_: round(ms / c, 4) for n_, (ms, c) in k.items()}, int(d_n.item()) if hits else "")
