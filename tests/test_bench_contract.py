"""bench.py prints ONE JSON line with the fields the measurement contract names (task statement, section 4):
run on the small C1 configuration so that the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "Mpaths/s" and d["unit"] == "Mpaths/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["scaling"] == "strong" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # value = paths of one step / time per step
    paths = d["config"]["width"] * d["config"]["height"] * d["config"]["spp"]
    assert abs(d["value"] - paths / d["ms_per_step"] / 1e3) <= 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "peak_calibrated", "unit", "frac", "traffic", "kernel", "kernel_ms",
                "launches_per_step", "lane_util", "hbm"):
        assert key in r, key
    # the roof that binds these kernels is vector-ALU issue; the HBM side is reported beside it
    # peak = the guide's figure (1024 SIMDs x 2.4 GHz / 2 cycles); the self-measured one is beside it, never instead
    assert r["bound"] == "valu" and r["unit"] == "Gwave-inst/s" and abs(r["peak"] - 1228.8) < 0.1
    assert r["peak_calibrated"] is None or r["peak_calibrated"] < r["peak"]
    for key in ("algorithmic_bytes_per_launch", "algorithmic_frac", "cache_served", "peak_GBps"):
        assert key in r["hbm"], key
    assert r["kernel"].startswith("wf_") and r["kernel_ms"] > 0
    # exclusive kernel times: the launches of a step add up to (at most) the step
    total = sum(k["ms_per_launch"] * k["launches_per_step"] for k in d["kernels"].values())
    assert total <= d["ms_per_step"] * 1.0001
    if r["achieved"] is not None:  # PMC figures available for this configuration
        # the headline fraction is the USEFUL one: issue slots whose lanes worked
        assert abs(r["issue_frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["issue_frac"] <= 1.0
        assert abs(r["frac"] - r["issue_frac"] * r["lane_util"]) < 1e-9 and 0.0 < r["frac"] <= r["issue_frac"]
        assert 0.0 < r["lane_util"] <= 1.0 and 0.0 <= r["hbm"]["measured_frac"] <= 1.0
        for k in d["kernels"].values():
            assert k.get("lane_util", 0.0) <= 1.0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
