"""bench.py as the driver runs it, on the one GPU of the test box: the default single-rank line (shape of the JSON contract,
`verified`, both roofline objects) and the sharded build + exchange path with the REAL RCCL backend at world size 1
(`--force-dist`: process-group initialisation on the device, the in-place `all_gather_into_tensor` on the grid's own bitmask
through the zero-copy tensor view, stream ordering between the library's launches and RCCL's) -- a multi-GPU node only the
driver's round-end run has."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_bench(*args):
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(free_port())   # the rendezvous of --force-dist: never a fixed port on a shared box
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract(gpu):
    d = run_bench("--steps", "5", "--warmup", "2", "--big-rays", "0", "--cpu-runs", "1", "--cpu-ray-sample", "200")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["verified"] is True
    assert d["metric"] == "Mrays/s" and d["value"] > 0 and d["vs_baseline"] is None and "workload" in d["config"]
    roof = d["roofline"]
    assert roof["kernel"] == "k_walk" and roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["achieved"] is not None and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4 and roof["traffic"]
    assert d["roofline_issue"]["bound"] == "valu_issue" and 0.0 < d["roofline_issue"]["frac"] < 1.0
    kv = d["kernel_rooflines"]["k_voxelize"]
    # (the tiled build mask of round 3 took the kernel from the memory side's request rate to VALU issue; both figures are on the line)
    assert kv["bound"] == "valu_issue" and 0.2 < kv["frac"] < 1.0 and 0.1 < kv["frac_of_request_rate"] < 1.2
    assert d["list_async"] is True
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]


def test_bench_rccl_path_world1(gpu):
    d = run_bench("--force-dist", "--dist-backend", "nccl", "--steps", "3", "--warmup", "1", "--big-rays", "0", "--no-cpu-baseline", "--c4-grid", "0",
                  "--rays", "200000")
    assert d["n_gpus"] == 1 and d["rccl_world"] == 1 and d["verified"] is True
    assert "nccl" in d["exchange"]["algo"] and "in place" in d["exchange"]["algo"]
    assert d["exchange"]["bytes_per_rank"] == 512 ** 3 // 8
