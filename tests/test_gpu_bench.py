"""bench.py's contract on the GPU box: ONE JSON line with the metric of BASELINE.json, the roofline object (dominant
kernel + one entry per kernel) and the CPU baseline (bounded sample).  Small step counts: this checks the shape of the
line, not the numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]            # exactly one line on stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("cfg", ["c3", "c5"])
def test_bench_line_contract(hiplib, cfg):
    extra = ["--pulses", "4"] if cfg == "c5" else []
    d = _bench("--config", cfg, "--steps", "2", "--warmup", "1", "--cpu-paths", "4096", *extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "Mrays/s" and d["value"] > 0 and d["ms_per_step"] > 0 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"].startswith("synthetic") and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernels"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    names = " ".join(k["kernel"] for k in r["kernels"])
    assert "wf_shade" in names and "wf_trace" in names and "bf_render_kernel" in names
    for k in r["kernels"]:
        assert k["avg_launch_ms"] >= 0 and 0 <= k["share_of_gpu_time"] <= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and "sample" in c


def test_bench_strong_scaling_line_at_one_gpu(hiplib):
    d = _bench("--config", "c4", "--scaling", "strong", "--steps", "2", "--warmup", "1", "--no-cpu", "--paths", "262144")
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["value"] > 0
