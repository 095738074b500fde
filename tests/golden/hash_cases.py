"""Prints, as one JSON object, the regression hashes of a few small cases computed with the oracle build named by
$VXORACLE_SO (see oracle/oracle.py).  Run as a subprocess by tests/test_oracle_compilers.py: one process per oracle build."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import oracle  # noqa: E402
import vx_scenes  # noqa: E402

CASES = [("cube", 0.25), ("cube", 0.0625), ("rotcube", 0.09), ("adversarial", 0.125), ("adversarial", 0.1), ("soup2000", 0.02), ("soup70000", 0.004)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


out = {}
for name, vs in CASES:
    v, t = vx_scenes.scene(name)
    vs = np.float32(vs)
    w, calls, gi = oracle.build_bool(v, t, vs)
    a = oracle.bool_aabbs(w, gi, vs)
    vec = oracle.build_vec(v, t, vs)
    oc = oracle.octree(v, t, vs)
    rays = vx_scenes.random_rays(2000, gi["bmin"], gi["bmax"], seed=11)
    tt, pp = oracle.trace_brute(a, rays)
    out["%s@%.9g" % (name, vs)] = dict(words_sha=sha(w), aabbs_sha=sha(a), vec_sha=sha(vec), octree_items_sha=sha(oc["items"]),
                                       octree_nodes_sha=sha(oc["nodes"]), set_calls=int(calls), trace_sha=sha(tt) + sha(pp))
print(json.dumps(out))
