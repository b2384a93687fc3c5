#!/usr/bin/env python3
"""Static VALU instruction counts of a 4-wide node visit (the product's) and of an 8-wide one (scripts/wide_node_cost.hip),
the product's compiler flags, gfx950 ISA.  Tooling for profiles/r03_wide_bvh_experiment.txt.

    python scripts/wide_node_cost.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from unit_costs import FLAGS, census  # noqa: E402


def main():
    with tempfile.TemporaryDirectory() as tmp:
        s = os.path.join(tmp, "wide.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-device-only", "-S", os.path.join(ROOT, "scripts", "wide_node_cost.hip"), "-o", s])
        kernels = census(open(s).read())
    base = sum(v for op, v in kernels["unit_baseline_ray"].items() if op.startswith("v_"))
    for name in ("unit_node4", "unit_node8_sorted", "unit_node8_mask", "unit_node8_pop"):
        ops = kernels[name]
        valu = sum(v for op, v in ops.items() if op.startswith("v_")) - base
        half = sum(v for op, v in ops.items() if op.startswith(("v_min", "v_max", "v_cndmask", "v_cmp", "v_cvt", "v_fma_mix", "v_pk_", "v_med3", "v_lshl", "v_bfe", "v_and_or", "v_ldexp")))
        vmem = sum(v for op, v in ops.items() if op.startswith(("global_", "buffer_", "flat_")))
        print("%-18s VALU %4d (of which half-rate kinds %4d)  vector-memory %2d" % (name[5:], valu, half, vmem))
    print("(the stack stores of these harnesses are never read back and compile away: the counts are without the pushes - at most "
          "three per 4-wide visit, seven per sorted 8-wide visit, one per masked one)")


if __name__ == "__main__":
    main()
