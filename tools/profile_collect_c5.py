"""Turns gpurun_out/prof_c5/ (tools/profile_c5.sh: rocprofv3 passes over tools/run_c5.py, BASELINE configs[4]) into
profiles/<tag>_c5_kernel_stats.csv, <tag>_c5_traffic.json and <tag>_c5_run.log."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_c5")
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
out = os.path.join(ROOT, "profiles")
ks = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(out, tag + "_c5_kernel_stats.csv"))
log = os.path.join(src, "kt.log")
if os.path.exists(log):
    with open(log) as fh, open(os.path.join(out, tag + "_c5_run.log"), "w") as oh:
        oh.write("".join(l for l in fh if l.startswith("[c5 ")))


def short(name):
    return name.replace("void ", "").split("(")[0]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE"):
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
traffic = {"workload": "tools/run_c5.py: 10M-triangle soup @ 2048^3, Bool build x2, Octree build x5, primary rays",
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per-launch averages, KB * 1024; FETCH_SIZE raw (gfx950 reports half of "
                   "the bytes of wide coalesced streaming reads: the x2 bound is given beside it)", "kernels": {}}
for k, v in sorted(acc.items()):
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 1024.0
        w = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024.0
        traffic["kernels"][k[:90]] = {"launches": len(v["FETCH_SIZE"]), "fetch_bytes_raw": int(f), "write_bytes": int(w), "hbm_bytes_per_launch": int(f + w),
                                      "hbm_bytes_if_fetch_x2": int(2 * f + w)}
json.dump(traffic, open(os.path.join(out, tag + "_c5_traffic.json"), "w"), indent=1)
for k, e in traffic["kernels"].items():
    print("%-90s %s" % (k, e))
