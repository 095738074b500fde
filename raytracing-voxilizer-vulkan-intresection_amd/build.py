"""Builds libvoxhip.so (HIP kernels + C ABI) for gfx950 in-tree, and the C++ facade's CLI / self-test binaries.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract: the
occupancy is bit-exact only if no multiply-add is fused.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
CPP = os.path.join(HERE, "cpp")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libvoxhip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
EXTRA = os.environ.get("VOXHIP_EXTRA_FLAGS", "").split()
# -fno-slp-vectorize: the SLP vectorizer packs the SAT / plane arithmetic into v_pk_*_f32 pairs and then spends as many v_mov
# instructions arranging register pairs as it saved (k_voxelize -4 %, k_trace -2 % without it); results are identical.
FLAGS = EXTRA + ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         "--offload-arch=" + ARCH]

# per-source flags.  vx_walk.hip: SimplifyCFG turns more two-sided branches of the ray kernel into selects (default threshold 4) -- 3-4 % of
# k_walk at 1M and at 8M rays, any value from 8 to 48; no effect on the other kernels (measured on vx_kernels.hip)
SOURCE_FLAGS = {"vx_walk.hip": ["-mllvm", "-two-entry-phi-node-folding-threshold=16"]}

SOURCES = ["vx_kernels.hip", "vx_trace.hip", "vx_walk.hip", "vx_octree.hip", "vx_sort.hip", "vx_api.cpp", "vx_obj.cpp", "vx_prof.cpp"]
HEADERS = ["vx_math.h", "vx_internal.h", os.path.join(ROOT, "include", "voxhip.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd[:3]))
    return r.stdout


def build_lib(force=False, verbose=False):
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    # a change of flags (diagnostic -D builds) invalidates every object
    stamp = os.path.join(objdir, "flags.txt")
    flags_now = " ".join(FLAGS) + " | " + repr(sorted(SOURCE_FLAGS.items()))
    flags_changed = not os.path.exists(stamp) or open(stamp).read() != flags_now
    # VOXHIP_VARIANT_ONLY=a.hip,b.hip: the extra -D flags only concern these sources (parameter sweeps of one kernel)
    only = [x for x in os.environ.get("VOXHIP_VARIANT_ONLY", "").split(",") if x]
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or (flags_changed and (not only or src in only)) or _newer(obj, [sp] + hdrs):
            cmd = [HIPCC] + FLAGS + SOURCE_FLAGS.get(src, []) + ["-x", "hip", "-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            _run(cmd)
    if force or _newer(LIB, objs):
        _run([HIPCC, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs)
    with open(stamp, "w") as fh:
        fh.write(flags_now)
    return LIB


def build_cpp(force=False):
    """C++ facade: the `voxilizer <obj> <voxelsize>` CLI and the facade self-test, plain g++ against libvoxhip.so."""
    outs = []
    for name in ("voxilizer", "facade_selftest"):
        src = os.path.join(CPP, name + ".cpp")
        if not os.path.exists(src):
            continue
        out = os.path.join(HERE, name)
        deps = [src, LIB] + [os.path.join(CPP, f) for f in os.listdir(CPP) if f.endswith((".hpp", ".h"))]
        if force or _newer(out, deps):
            _run(["g++", "-O2", "-std=c++20", "-ffp-contract=off", "-I", CPP, "-I", os.path.join(ROOT, "include"), src, "-o", out,
                  "-L", HERE, "-lvoxhip", "-Wl,-rpath,$ORIGIN", "-lpthread"])
        outs.append(out)
    return outs


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_lib(force=force, verbose=True))
    for o in build_cpp(force=force):
        print(o)
