#!/usr/bin/env python3
"""rocprofv3 PMC passes over the wavefront kernels, and their summary (measurement tooling, not product).

Three passes per configuration, each a child process `rocprofv3 --pmc ... -- python3 scripts/quick_bench.py --config <c>
--single-chain --steps 1` (whole-batch launches = the launch shape of the pipelined default; PMC collection serialises
the dispatches, so every launch runs alone):

    sq     SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS
           SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
    fetch  FETCH_SIZE            (TCC: 3 of the 4 slots - MI355X_MICROARCH.md "rocprofv3 PMC slots")
    write  WRITE_SIZE

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB): the guide's gfx950 correction (FETCH_SIZE tallies a 128-B
request of a 16 B/lane stream at 64 B).  lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU).

    python scripts/pmc_tools.py c2 [c3 c4 ...]      -> profiles/r04_<config>_pmc.json  (run on the GPU box)
bench.py calls collect() itself for the headline configuration, so the figures in its line are measured in that run.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SQ_COUNTERS = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM",
               "SQ_INSTS_LDS", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"]
PASSES = {"sq": SQ_COUNTERS, "fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"]}
# optional fourth pass (profiles only): what kinds of vector instructions a kernel issues.  ADD / MUL / FMA F32 are the
# full-rate class of profiles/r02_valu_calibration.json; conversions and transcendentals are slower; the rest of
# SQ_INSTS_VALU is compares, selects, min / max, moves, shifts and integer arithmetic
MIX_COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32",
                "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT"]


def kernel_source_hash():
    """sha256 over the kernel and host sources of the library: a committed profile is only quoted for the code it
    was taken on."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, "prosper_amd", "csrc")
    files = []
    for dirpath, _, names in os.walk(base):
        for n in names:
            if n.endswith((".hip", ".hpp", ".cpp")) or n == "Makefile":
                files.append(os.path.join(dirpath, n))
    for f in sorted(files):
        h.update(os.path.relpath(f, base).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _base_name(kernel_name):
    n = kernel_name.split("(")[0].replace("void ", "").replace("ppt::", "")
    counted = "<true" in n
    return n.split("<")[0], counted


def _rows(directory):
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def summarise(directory):
    """-> {kernel: {counter: mean per launch, 'launches': n, 'us_per_launch': mean duration under the profiler}}"""
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in _rows(directory):
        name, counted = _base_name(r["Kernel_Name"])
        if counted or not name.startswith("wf_"):
            continue
        d = per[name][r["Dispatch_Id"]]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        try:
            d["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        except (KeyError, ValueError):
            pass
    out = {}
    for name, disp in per.items():
        n = len(disp)
        mean = collections.defaultdict(float)
        for d in disp.values():
            for k, v in d.items():
                mean[k] += v / n
        m = dict(mean)
        m["launches"] = n
        m["us_per_launch"] = m.pop("_ns", 0.0) / 1e3
        out[name] = m
    return out


def run_pass(config, counters, out_dir, extra_args=(), timeout=600):
    """One rocprofv3 --pmc child process; raises on failure."""
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    cmd = ["rocprofv3", "--pmc"] + list(counters) + ["-d", out_dir, "-o", "pmc", "--output-format", "csv", "--",
                                                    sys.executable, os.path.join(ROOT, "scripts", "quick_bench.py"),
                                                    "--config", config, "--single-chain", "--steps", "1"] + list(extra_args)
    # its own process group: a pass that hangs (some counter sets do) is killed as a whole - profiler AND the profiled
    # python child - before the caller goes on to its timed region; nothing of it may be left holding the GPU
    p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.wait()
        raise RuntimeError("rocprofv3 pass timed out after %d s (its process group was killed)" % timeout)
    if p.returncode != 0:
        raise RuntimeError("rocprofv3 pass failed (%d): %s" % (p.returncode, out[-800:]))


def collect(config, work_dir=None, extra_args=(), keep=False, timeout=600, with_mix=False):
    """Runs the three passes and returns the per-kernel summary dict (see module docstring)."""
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 is not on PATH")
    own = work_dir is None
    if own:
        work_dir = tempfile.mkdtemp(prefix="prosper_pmc_", dir="/tmp")
    try:
        parts = {}
        for name, counters in PASSES.items():
            d = os.path.join(work_dir, "%s_%s" % (config, name))
            run_pass(config, counters, d, extra_args, timeout)
            parts[name] = summarise(d)
        mix = {}
        if with_mix:
            d = os.path.join(work_dir, "%s_mix" % config)
            run_pass(config, MIX_COUNTERS, d, extra_args, timeout)
            mix = summarise(d)
        kernels = {}
        for k, sq in parts["sq"].items():
            valu = sq.get("SQ_INSTS_VALU", 0.0)
            gui = sq.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
            fetch_kb = parts["fetch"].get(k, {}).get("FETCH_SIZE", 0.0)
            write_kb = parts["write"].get(k, {}).get("WRITE_SIZE", 0.0)
            kernels[k] = {
                "launches_profiled": sq["launches"],
                "valu_insts_per_launch": valu,
                "salu_insts_per_launch": sq.get("SQ_INSTS_SALU", 0.0),
                "vmem_insts_per_launch": sq.get("SQ_INSTS_VMEM", 0.0),
                "lds_insts_per_launch": sq.get("SQ_INSTS_LDS", 0.0),
                "waves_per_launch": sq.get("SQ_WAVES", 0.0),
                "lane_util": (sq.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * valu)) if valu else 0.0,
                "cu_busy": (sq.get("SQ_BUSY_CU_CYCLES", 0.0) / (256.0 * gui)) if gui else 0.0,
                "cycles_per_valu_inst_per_simd_profiled": (1024.0 * gui / valu) if valu else 0.0,
                "us_per_launch_profiled": sq.get("us_per_launch", 0.0),
                "ghz_profiled": (gui / (sq["us_per_launch"] * 1e3)) if sq.get("us_per_launch") else 0.0,
                "FETCH_SIZE_KB_per_launch": fetch_kb,
                "WRITE_SIZE_KB_per_launch": write_kb,
                "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
            }
            if k in mix and mix[k].get("SQ_INSTS_VALU"):
                total = mix[k]["SQ_INSTS_VALU"]
                named = {n[len("SQ_INSTS_VALU_"):].lower(): mix[k].get(n, 0.0) / total for n in MIX_COUNTERS[1:]}
                named["other"] = 1.0 - sum(named.values())
                kernels[k]["valu_mix"] = named
        return {"config": config, "kernel_source_sha16": kernel_source_hash(),
                "note": "rocprofv3 --pmc, three passes (SQ + GRBM; FETCH_SIZE; WRITE_SIZE) over scripts/quick_bench.py "
                        "--single-chain --steps 1: whole-batch launches, each alone on the GPU; means per launch; "
                        "hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction of MI355X_MICROARCH.md)",
                "kernels": kernels}
    finally:
        if own and not keep:
            shutil.rmtree(work_dir, ignore_errors=True)


def load_committed(config):
    """profiles/r04_<config>_pmc.json if it was taken on the current kernel sources, else None."""
    path = os.path.join(ROOT, "profiles", "r04_%s_pmc.json" % config)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if d.get("kernel_source_sha16") != kernel_source_hash():
        return None
    d["source"] = "profiles/r04_%s_pmc.json (taken on these kernel sources: sha16 %s)" % (config, d["kernel_source_sha16"])
    return d


def main():
    for config in sys.argv[1:]:
        extra = []
        out = collect(config, extra_args=extra, with_mix=True)
        path = os.path.join(ROOT, "profiles", "r04_%s_pmc.json" % config)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")
        for k, v in sorted(out["kernels"].items()):
            print("%s %-20s launches %3d  VALU %.3e  lane_util %.2f  cyc/inst/simd %.2f  cu_busy %.2f  HBM %.3f GB  %.0f us" % (
                config, k, v["launches_profiled"], v["valu_insts_per_launch"], v["lane_util"],
                v["cycles_per_valu_inst_per_simd_profiled"], v["cu_busy"], v["hbm_bytes_per_launch"] / 1e9,
                v["us_per_launch_profiled"]))
            if "valu_mix" in v:
                print("      mix: " + "  ".join("%s %.2f" % kv for kv in v["valu_mix"].items()))


if __name__ == "__main__":
    main()
