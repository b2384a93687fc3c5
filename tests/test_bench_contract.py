"""bench.py prints ONE JSON line - under 4 KB, so that every tail the driver keeps holds all of it - with the fields the
measurement contract names (task statement, section 4), and writes the per-kernel detail to bench_detail.json: run on the
small C1 configuration so that the test takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_slim_line_of_a_full_result_stays_under_4_kb():
    """The round-3 line (31 KB: five sub-configurations with full roofline / kernel / counter objects) through slim_line."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench.json")))
    # round 4's extra sub-configurations, with every field a sub-configuration can carry
    full["configs"]["c1"] = dict(full["configs"]["c3"])
    full["configs"]["c5_rank_share"] = {"ms_per_step": 49.2, "Mpaths_per_s": 1350.0, "extrapolated": "x" * 90,
                                        "cpu_baseline": {"value": 0.51}}
    for c in full["configs"].values():
        c.setdefault("cpu_baseline", {"value": 0.52})
    full["ranks_seen"], full["gather_ms"] = 8, 0.123
    text = bench.slim_line(full)
    assert len(text) < bench.SLIM_LIMIT == 4096 and "\n" not in text
    d = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "configs", "detail"):
        assert key in d, key
    assert set(d["configs"]) == {"c1", "c3", "c4", "helmet", "helmet2k", "c5_rank_share"}
    assert d["configs"]["c3"]["cpu_Mpaths_per_s"] == 0.52 and d["configs"]["c4"]["frac"] > 0
    assert d["roofline"]["hbm"]["cache_served"] is True and isinstance(d["roofline"]["frac"], float)
    # a result that cannot fit loses its optional parts, never its contract fields
    full["configs"] = {"c%d" % i: dict(full["configs"]["c3"], extrapolated="y" * 200) for i in range(40)}
    text = bench.slim_line(full)
    assert len(text) < 4096 and "roofline" in json.loads(text) and "cpu_baseline" in json.loads(text)


@pytest.mark.gpu
def test_bench_json_line_has_the_contract_fields(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1", "--steps", "5", "--warmup", "2",
                          "--detail", str(tmp_path / "detail.json")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    assert len(lines[0]) < 4096, "the line must fit every tail the driver keeps"
    d = json.loads(lines[0])
    detail = json.load(open(d["detail"]))
    assert detail["value"] == pytest.approx(d["value"], rel=1e-4) and "kernels" in detail and "counters" in detail
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
    assert abs(d["value"] - paths / d["ms_per_step"] / 1e3) <= 1e-4 * d["value"]  # (the line carries 5 significant digits)
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "peak_calibrated", "unit", "frac", "traffic", "kernel", "kernel_ms",
                "launches_per_step", "lane_util", "hbm", "issue_frac"):
        assert key in r, key
    # the roof that binds these kernels is vector-ALU issue; the HBM side is reported beside it
    # peak = the guide's figure (1024 SIMDs x 2.4 GHz / 2 cycles); the self-measured one is beside it, never instead
    assert r["bound"] == "valu" and r["unit"] == "Gwave-inst/s" and abs(r["peak"] - 1228.8) < 0.1
    assert r["peak_calibrated"] is None or r["peak_calibrated"] < r["peak"]
    for key in ("algorithmic_bytes_per_launch", "algorithmic_frac", "cache_served", "peak_GBps"):
        assert key in r["hbm"], key
    assert r["kernel"].startswith("wf_") and r["kernel_ms"] > 0
    # exclusive kernel times: the launches of a step add up to (at most) the step
    total = sum(k["ms_per_launch"] * k["launches_per_step"] for k in detail["kernels"].values())
    assert total <= detail["ms_per_step"] * 1.0001
    if r["achieved"] is not None:  # PMC figures available for this configuration
        # the headline fraction is the USEFUL one: issue slots whose lanes worked
        assert abs(r["issue_frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0.0 < r["issue_frac"] <= 1.0
        assert abs(r["frac"] - r["issue_frac"] * r["lane_util"]) < 1e-4 and 0.0 < r["frac"] <= r["issue_frac"]
        assert 0.0 < r["lane_util"] <= 1.0 and 0.0 <= r["hbm"]["measured_frac"] <= 1.0
        for k in detail["kernels"].values():
            assert k.get("lane_util", 0.0) <= 1.0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1


@pytest.mark.gpu
def test_plain_invocation_with_several_gpus_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts torch.distributed.run as a child
    process before anything touches the GPU and relays rank 0's line.  On a one-GPU box the two ranks share GPU 0
    (PROSPER_BENCH_REHEARSE=1: tiles through host memory over gloo, the product's de-interleave kernel on the root) -
    a rehearsal of the flow, never a measurement: the gathered image must be the one-GPU image."""
    env = dict(os.environ, PROSPER_BENCH_REHEARSE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--config", "c1", "--steps", "3", "--warmup", "1", "--no-pmc", "--no-cpu-baseline", "--no-subconfigs", "--no-extras",
              "--detail", str(tmp_path / "detail.json")]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d1, d2 = json.loads(one.stdout.strip().splitlines()[-1]), json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["ranks_seen"] == 2 and d1["n_gpus"] == 1
    assert "REHEARSAL" in d2["config"]["parallelism"]
    assert d2["mean_radiance"] == d1["mean_radiance"] and d2["mean_radiance"] > 0
    # a rank count that does not divide the stripes is refused before any rank starts
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"] + common, capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=env)
    assert bad.returncode != 0 and "do not divide" in bad.stderr
