# diagnostic: needs a library built with VOXHIP_EXTRA_FLAGS=-DVX_TRACE_DEBUG_UTIL (per-phase wave cycles of k_trace)
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import numpy as np, torch, voxhip, vx_scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
v, t = vx_scenes.scene("atrium262k")
g = voxhip.Grid.voxelize(voxhip.Mesh.from_arrays(v, t), np.float32(32 / 512))
rays = torch.from_numpy(vx_scenes.random_rays(n, v.min(0), v.max(0), seed=2)).cuda()
d_t = torch.empty(n, dtype=torch.float32, device="cuda"); d_p = torch.empty(n, dtype=torch.int32, device="cuda")
g.trace_device(rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr()); torch.cuda.synchronize()
L = voxhip.lib()
out = (C.c_ulonglong * 24)()
L.vx_debug_trace_util(out, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
g.trace_device(rays.data_ptr(), n, d_t.data_ptr(), d_p.data_ptr())
e1.record(); torch.cuda.synchronize()
print("trace_device ms %.3f" % e0.elapsed_time(e1))
L.vx_debug_trace_util(out, 0)
o = list(out)
tot = sum(o[:5])
names = ["refill", "donate", "walk", "brick", "retire"]
print("rays", n, "wave cycles total %.3g" % tot)
for i in range(5):
    print("  %-7s %5.1f %%" % (names[i], 100.0 * o[i] / tot))
print("walk iterations %d, mean active lanes %.1f" % (o[5], o[6] / max(o[5], 1)))
print("brick phases    %d, mean lanes with a brick %.1f" % (o[7], o[8] / max(o[7], 1)))
print("rounds          %d, mean busy lanes %.1f" % (o[9], o[10] / max(o[9], 1)))
print("cycles per walk iteration %.0f, per brick phase %.0f" % (o[2] / max(o[5], 1), o[3] / max(o[7], 1)))
print("brick tests: %d past culling of %d; per test past culling: %.2f slices, %.2f rows, %.2f slab tests" % (o[19], o[8], o[16] / max(o[19], 1), o[17] / max(o[19], 1), o[18] / max(o[19], 1)))
print("brick phase divergence: slowest lane's work / mean work of the lanes with a brick = %.2f" % (o[11] / max(o[12] / max(o[8] / max(o[7], 1), 1), 1)))
print("waves %d: mean lifetime %.3g cycles, longest %.3g (x%.2f); mean time after the wave's queue ran dry %.3g cycles (%.0f %% of its life)" % (o[22], tot / max(o[22], 1), o[20], o[20] / max(tot / max(o[22], 1), 1), o[21] / max(o[22], 1), 100.0 * o[21] / max(tot, 1)))
