#!/usr/bin/env python3
"""Turns rocprofv3 --pmc passes (one directory per config) into profiles/r01_valu_issue.{json,txt}.

    python scripts/make_valu_issue.py out_prefix c2=<dir> c3=<dir> c4=<dir>

Counters: SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SALU
SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE.  valu_busy = SQ_ACTIVE_INST_VALU / (256 CUs * GRBM_GUI_ACTIVE / 8 XCDs),
lane_util = SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU), cu_busy = SQ_BUSY_CU_CYCLES / (256 * GUI / 8).
"""
import collections
import csv
import glob
import json
import sys

NOTE = ("rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_INSTS_SALU "
        "SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE over scripts/quick_bench.py --config <c> --small-textures --single-chain "
        "--steps 1: whole-batch launches, the launch shape of the default pipelined mode (PMC collection serialises the "
        "dispatches, so each launch runs alone); valu_busy = "
        "SQ_ACTIVE_INST_VALU / (256 CUs * GRBM_GUI_ACTIVE/8 XCDs) (quad-cycles per SIMD-quad), lane_util = "
        "SQ_THREAD_CYCLES_VALU / (64*SQ_INSTS_VALU); mean per launch")


def summarise(directory):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for f in glob.glob(directory + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ppt::", "").split("<")[0]
            if not name.startswith("wf_"):
                continue
            d = per[name][r["Dispatch_Id"]]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {}
    for name, disp in sorted(per.items()):
        n = len(disp)
        mean = collections.defaultdict(float)
        for d in disp.values():
            for k, v in d.items():
                mean[k] += v / n
        gui = mean["GRBM_GUI_ACTIVE"] / 8.0
        valu = mean["SQ_INSTS_VALU"]
        out[name] = {"launches": n, "valu_insts_per_launch": valu,
                     "lane_util": round(mean["SQ_THREAD_CYCLES_VALU"] / (64.0 * valu), 3) if valu else 0.0,
                     "valu_busy": round(mean["SQ_ACTIVE_INST_VALU"] / (256.0 * gui), 3) if gui else 0.0,
                     "cu_busy": round(mean["SQ_BUSY_CU_CYCLES"] / (256.0 * gui), 3) if gui else 0.0}
    return out


def main():
    prefix = sys.argv[1]
    configs = {}
    for arg in sys.argv[2:]:
        c, d = arg.split("=")
        configs[c] = summarise(d)
    json.dump({"note": NOTE, "configs": {c: {k: {a: b for a, b in v.items() if a != "launches"} for k, v in ks.items()}
                                         for c, ks in configs.items()}}, open(prefix + ".json", "w"), indent=1)
    with open(prefix + ".txt", "w") as f:
        f.write("# " + NOTE + "\n")
        for c, ks in configs.items():
            for k, v in ks.items():
                f.write("%s %-20s launches %2d  VALU insts %.3e  lane_util %.2f  valu_busy %.2f  cu_busy %.2f\n" % (
                    c, k, v["launches"], v["valu_insts_per_launch"], v["lane_util"], v["valu_busy"], v["cu_busy"]))


if __name__ == "__main__":
    main()
