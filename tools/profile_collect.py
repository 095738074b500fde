"""Turns gpurun_out/prof_round/ (tools/profile_round.sh) into the committed summaries under profiles/:
   <tag>_kernel_stats.csv, <tag>_pmc_raw.txt, <tag>_traffic.json (read by bench.py for roofline.traffic)."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_round")
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
out = os.path.join(ROOT, "profiles")

ks = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(out, tag + "_kernel_stats.csv"))


def short(name):
    n = name.replace("void ", "")
    n = n.split("(")[0]
    return n


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


lines = []
allc = collections.defaultdict(dict)
for d in sorted(os.listdir(src)):
    if not d.startswith("pmc_") or not os.path.isdir(os.path.join(src, d)):
        continue
    for k, v in sorted(counters(d).items()):
        for c, x in sorted(v.items()):
            allc[k][c] = sum(x) / len(x)
        lines.append("%-44s %s n=%d" % (k[:44], " ".join("%s=%.5g" % (c, sum(x) / len(x)) for c, x in sorted(v.items())), len(next(iter(v.values())))))
open(os.path.join(out, tag + "_pmc_raw.txt"), "w").write("\n".join(lines) + "\n")

traffic = {"workload": "atrium262k", "grid": 512, "rays": 1000000,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (TCC slots), per-launch averages, KB*1024. "
                   "gfx950 caveat (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads exactly 1/2 of the bytes of a wide (16 B/lane) coalesced "
                   "streaming read; the accesses of these kernels are 4-8 B/lane gathers and 24 B strided records, which the guide calls "
                   "uncalibrated, so the raw figure is reported and the x2 bound is given beside it.",
           "kernels": {}}
for k, v in allc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and k.startswith("vx::"):
        base = k[4:].split("<")[0]
        f, w = v["FETCH_SIZE"] * 1024.0, v["WRITE_SIZE"] * 1024.0
        e = traffic["kernels"].setdefault(base, {"fetch_bytes_raw": 0, "write_bytes": 0, "variants": 0})
        e["fetch_bytes_raw"] += f
        e["write_bytes"] += w
        e["variants"] += 1
for base, e in traffic["kernels"].items():
    n = e.pop("variants")
    e["fetch_bytes_raw"] = int(e["fetch_bytes_raw"] / n)
    e["write_bytes"] = int(e["write_bytes"] / n)
    e["hbm_bytes_per_launch"] = e["fetch_bytes_raw"] + e["write_bytes"]
    e["hbm_bytes_if_fetch_x2"] = 2 * e["fetch_bytes_raw"] + e["write_bytes"]
json.dump(traffic, open(os.path.join(out, tag + "_traffic.json"), "w"), indent=1)
# instruction-issue view of the same passes (the ray and SAT kernels are issue/latency bound, not HBM bound)
issue = {"note": "per launch, from the SQ_* passes: VALU wave-instructions, lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU), "
                 "share of wave cycles spent waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES", "kernels": {}}
for k, v in allc.items():
    if k.startswith("vx::") and "SQ_INSTS_VALU" in v and "SQ_ACTIVE_INST_VALU" in v and v["SQ_ACTIVE_INST_VALU"] > 0:
        base = k[4:].split("<")[0]
        if base in issue["kernels"]:
            continue
        e = {"valu_wave_insts": v["SQ_INSTS_VALU"], "salu_insts": v.get("SQ_INSTS_SALU"),
             "lane_utilisation": round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4)}
        if v.get("SQ_WAVE_CYCLES"):
            e["wait_share_of_wave_cycles"] = round(v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"], 4)
        issue["kernels"][base] = e
json.dump(issue, open(os.path.join(out, tag + "_issue.json"), "w"), indent=1)
b = os.path.join(src, "bench_under_kernel_trace.json")
if os.path.exists(b) and os.path.getsize(b):
    shutil.copy(b, os.path.join(out, tag + "_bench_under_kernel_trace.json"))
print("\n".join(lines))
