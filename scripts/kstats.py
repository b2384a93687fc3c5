#!/usr/bin/env python3
"""Print name/calls/avg-us from a rocprofv3 kernel_stats.csv directory."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0].replace("void ppt::", "")
        if "rocclr" in n or "triangles" in n: continue
        print("%-30s calls %3s avg %9.1f us total %9.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
