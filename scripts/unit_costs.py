#!/usr/bin/env python3
"""Static VALU instruction counts of one unit of path-tracing work each, from the gfx950 ISA of scripts/unit_costs.hip
(the product's device functions, the product's compiler flags): the yardstick bench.py prices the work counters with.

    python scripts/unit_costs.py [--out profiles/r04_unit_costs.json]

A unit's cost = VALU instructions of its kernel minus those of unit_baseline_ray (the same ray load and result store
without the work).  Both sides of every branch are counted."""
import argparse
import collections
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "scripts", "unit_costs.hip")
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-Wno-unused-function", "--offload-arch=gfx950"]


def census(asm):
    out, name, ops = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(unit_\w+):", line)
        if m:
            name, ops = m.group(1), collections.Counter()
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            out[name] = ops
            name = None
            continue
        if not t or t.startswith((".", ";", "/")) or t.endswith(":"):
            continue
        ops[t.split()[0]] += 1
    return out


def kernel_source_hash():
    h = hashlib.sha256()
    for f in sorted(os.listdir(os.path.join(ROOT, "prosper_amd", "csrc"))):
        if f.endswith((".hpp", ".hip")):
            h.update(open(os.path.join(ROOT, "prosper_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "unit_costs.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-device-only", "-S", SRC, "-o", s])
        kernels = census(open(s).read())
    valu = {k: sum(v for op, v in ops.items() if op.startswith("v_")) for k, ops in kernels.items()}
    vmem = {k: sum(v for op, v in ops.items() if op.startswith(("global_", "buffer_", "flat_", "scratch_"))) for k, ops in kernels.items()}
    base = valu["unit_baseline_ray"]
    units = {}
    for k in sorted(valu):
        if k == "unit_baseline_ray":
            continue
        units[k[len("unit_"):]] = {"valu": valu[k] - base, "valu_raw": valu[k], "vmem": vmem[k], "lds": sum(
            v for op, v in kernels[k].items() if op.startswith("ds_"))}
    units["triangle_finish"] = {"valu": units["triangle_full"]["valu"] - units["triangle_edge_functions"]["valu"],
                                "note": "triangle_full - triangle_edge_functions: distance, range, box guard, barycentrics"}
    units["any_hit_exact"] = {"valu": units["any_hit_with_texels"]["valu"] - units["any_hit_settle"]["valu"],
                              "note": "any_hit_with_texels - any_hit_settle: four taps, filter, sRGBtoLinear, comparison"}
    result = {"source": "scripts/unit_costs.hip through hipcc " + " ".join(FLAGS) + " -S, v_* instructions per kernel minus unit_baseline_ray (%d)" % base,
              "kernel_source_sha16": kernel_source_hash(), "units": units}
    text = json.dumps(result, indent=1)
    if args.out:
        open(args.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
