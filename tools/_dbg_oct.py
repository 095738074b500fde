import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, voxhip, vx_scenes, oracle
for NT, G in ((20000, 256), (200000, 512), (1000000, 1024)):
    v, t = vx_scenes.soup(NT, seed=4, edge=1.5 / G); vs = np.float32(1.0 / G)
    mesh = voxhip.Mesh.from_arrays(v, t)
    o = voxhip.Octree(mesh, vs)
    items = o.items(); nodes = o.nodes()
    bits = int(np.ceil(np.log2(G)))
    ref = oracle.octree_nodes_from_sorted_items(items, bits, 16)
    a = np.frombuffer(nodes.tobytes(), np.uint32).reshape(-1, 10); b = np.frombuffer(ref.tobytes(), np.uint32).reshape(-1, 10)
    print(NT, G, len(items), a.shape, b.shape, "equal" if a.shape == b.shape and np.array_equal(a, b) else "DIFF")
    if a.shape == b.shape and not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        print("  ndiff", len(bad), "first", bad[:8].tolist())
        for r, c in bad[:4]:
            print("  node", r, "col", c, "gpu", a[r].tolist(), "ref", b[r].tolist())
            ch = b[r, c] if c < 8 else None
            if ch is not None and ch != 0xFFFFFFFF:
                print("    expected child", ch, "ref child node", b[ch].tolist(), "gpu child node", a[ch].tolist())
