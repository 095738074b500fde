# experiment: does ray coherence (host-side sort by entry cell + direction octant) speed up k_trace?
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, voxhip, vx_scenes
v, t = vx_scenes.scene("atrium262k")
g = voxhip.Grid.voxelize(voxhip.Mesh.from_arrays(v, t), np.float32(32 / 512))
R = 1_000_000
rays = vx_scenes.random_rays(R, v.min(0), v.max(0), seed=2)
def run(r, label):
    d = torch.from_numpy(np.ascontiguousarray(r)).cuda(); dt = torch.empty(R, dtype=torch.float32, device="cuda"); dp = torch.empty(R, dtype=torch.int32, device="cuda")
    g.trace_device(d.data_ptr(), R, dt.data_ptr(), dp.data_ptr()); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.trace_device(d.data_ptr(), R, dt.data_ptr(), dp.data_ptr())
    e1.record(); torch.cuda.synchronize()
    print(label, "%.3f ms" % (e0.elapsed_time(e1) / 5), "hits", int((dt > 0).sum()))
run(rays, "unsorted")
# entry point into the bbox
o, d = rays[:, :3].astype(np.float64), rays[:, 3:].astype(np.float64)
lo, hi = v.min(0).astype(np.float64), v.max(0).astype(np.float64)
t1, t2 = (lo - o) / d, (hi - o) / d
tn = np.minimum(t1, t2).max(1).clip(0)
e = o + tn[:, None] * d
for bits in (3, 5):
    q = np.clip(((e - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    octant = (d[:, 0] < 0) * 1 + (d[:, 1] < 0) * 2 + (d[:, 2] < 0) * 4
    key = octant
    for b in range(bits):
        for a in range(3):
            key = key * 2 + ((q[:, a] >> (bits - 1 - b)) & 1)
    order = np.argsort(key, kind="stable")
    run(rays[order], "sorted by octant + %d-bit entry morton" % bits)
# sort by direction only
dirkey = np.clip(((d + 1) / 2 * 16).astype(np.int64), 0, 15)
order = np.lexsort((dirkey[:, 2], dirkey[:, 1], dirkey[:, 0]))
run(rays[order], "sorted by direction (16^3 bins)")
# sort by the step count proxy: in-grid chord length
t_out = np.maximum(t1, t2).min(1)
order = np.argsort(t_out - tn)
run(rays[order], "sorted by chord length")
