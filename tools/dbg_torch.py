import faulthandler, sys, os, time
faulthandler.dump_traceback_later(50, exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
def log(*a):
    print("%.2f" % time.time(), *a, flush=True)
order = sys.argv[1] if len(sys.argv) > 1 else "vox_first"
import numpy as np
if order == "torch_first":
    import torch; log("torch imported first", torch.cuda.is_available())
import voxhip, vx_scenes
log("devices", voxhip.device_count())
v, t = vx_scenes.rotated_cube()
mesh = voxhip.Mesh.from_arrays(v, t)
g = voxhip.Grid.voxelize(mesh, 0.09)
log("voxelized", g.describe()["occupied"])
import torch
log("torch imported", torch.__version__)
log("cuda available", torch.cuda.is_available())
rays = vx_scenes.random_rays(5000, v.min(0), v.max(0), seed=3)
dr = torch.from_numpy(rays).cuda(); log("rays on device")
dt = torch.empty(len(rays), dtype=torch.float32, device="cuda")
dp = torch.empty(len(rays), dtype=torch.int32, device="cuda")
dh = torch.zeros(len(rays) * 3, dtype=torch.int32, device="cuda")
dn = torch.zeros(1, dtype=torch.int64, device="cuda")
torch.cuda.synchronize(); log("sync1")
g.trace_device(dr.data_ptr(), len(rays), dt.data_ptr(), dp.data_ptr()); log("launched no-compaction")
torch.cuda.synchronize(); log("sync2", float((dt > 0).sum()))
g.trace_device(dr.data_ptr(), len(rays), dt.data_ptr(), dp.data_ptr(), dh.data_ptr(), dn.data_ptr()); log("launched compaction")
torch.cuda.synchronize(); log("sync3", int(dn.item()))
