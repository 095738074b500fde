import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, voxhip, vx_scenes, oracle
NT, G = 20000, 256
v, t = vx_scenes.soup(NT, seed=4, edge=1.5 / G); vs = np.float32(1.0 / G)
o = voxhip.Octree(voxhip.Mesh.from_arrays(v, t), vs)
items = o.items(); nodes = o.nodes()
ref = oracle.octree_nodes_from_sorted_items(items, 8, 16)
b = np.frombuffer(ref.tobytes(), np.uint32).reshape(-1, 10)
a = np.frombuffer(nodes.tobytes(), np.uint32).reshape(-1, 10)
n = len(items); cnt = np.bincount(b[:, 8], minlength=n); base = np.concatenate([[0], np.cumsum(cnt)])
i = 65925
print("ref: base[i]", base[i], "cnt", cnt[i], "node", b[base[i]].tolist(), "items", hex(items[i]), hex(items[i-1]))
for k in (65778, 65777, 65765):
    print(" pos", k, "base", base[k], "cnt", cnt[k], "item", hex(items[k]), hex(items[k-1]))
print("equal" if np.array_equal(a, b) else "DIFF %d" % len(np.argwhere(a != b)))
