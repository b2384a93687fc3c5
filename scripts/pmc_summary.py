#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ppt::", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    if "rocclr" in name or "triangles" in name:
        continue
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    n = len(next(iter(cs.values())))
    valu = m.get("SQ_INSTS_VALU", 0)
    util = m.get("SQ_THREAD_CYCLES_VALU", 0) / (valu * 64) if valu else 0
    print("%-28s n=%2d waves %7.0f VALU %.3e SALU %.3e VMEM %.3e LDS %.3e lane-util %.2f wave-cycles %.3e wait-any %.2f" % (
        name, n, m.get("SQ_WAVES", 0), valu, m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_VMEM", 0), m.get("SQ_INSTS_LDS", 0),
        util, m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_WAIT_ANY", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1))))
