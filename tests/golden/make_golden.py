"""Regenerates tests/golden/regression.json from the CPU oracle (oracle/vx_oracle.c).

These are SELF-GENERATED regression vectors: they freeze what the oracle produced when it was checked against the
survey anchors (survey_anchors.json), so that later edits to the oracle or the scene generators cannot drift silently.
They do not pin the oracle to the reference (nothing in the reference can: it ships no fixtures and cannot be built
here without stand-in headers).  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "raytracing-voxilizer-vulkan-intresection_amd")]
import oracle  # noqa: E402
import vx_scenes  # noqa: E402

CASES = [("cube", 0.25), ("cube", 0.0625), ("cube", 0.3), ("rotcube", 0.09), ("rotcube", 0.031), ("adversarial", 0.125),
         ("adversarial", 0.1), ("adversarial", 0.03125), ("soup2000", 0.02), ("blob70k", 2.0 / 64), ("blob70k", 2.0 / 256),
         ("atrium262k", 32.0 / 128)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {}
    for name, vs in CASES:
        v, t = vx_scenes.scene(name)
        vs = np.float32(vs)
        w, calls, gi = oracle.build_bool(v, t, vs)
        a = oracle.bool_aabbs(w, gi, vs)
        vec = oracle.build_vec(v, t, vs)
        oc = oracle.octree(v, t, vs)
        out["%s@%.9g" % (name, vs)] = dict(
            verts_sha=sha(v), tris_sha=sha(t), dim=list(gi["dim"]), occupied=int(len(a)), set_calls=int(calls),
            words_sha=sha(w), aabbs_sha=sha(a), vec_sha=sha(vec), octree_items_sha=sha(oc["items"]),
            octree_nodes=int(len(oc["nodes"])), octree_nodes_sha=sha(oc["nodes"]), octree_bytes=int(oc["bytes"]),
            first_aabb=[float(x) for x in (list(a[0]["mn"]) + list(a[0]["mx"]))] if len(a) else [])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "regression.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
