#!/usr/bin/env python3
"""Ad-hoc PMC comparison of trace-kernel variants (tooling): python scripts/pmc_ab.py <counters,comma> <config>..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_tools, tempfile
counters = sys.argv[1].split(",")
# FETCH_SIZE / WRITE_SIZE each take 3 of the 4 TCC slots (MI355X_MICROARCH.md): together, or next to the TCP latency counters, the
# profiler hangs until the box's watchdog kills the run
if sum(c in ("FETCH_SIZE", "WRITE_SIZE") for c in counters) > 1 or any("LATENCY" in c for c in counters):
    sys.exit("pmc_ab.py: one of FETCH_SIZE / WRITE_SIZE per pass, and no *_LATENCY counters")
for cfg in sys.argv[2:]:
    d = tempfile.mkdtemp(prefix="pmcab_", dir="/tmp")
    pmc_tools.run_pass(cfg, counters, d)
    for k, v in sorted(pmc_tools.summarise(d).items()):
        if "trace" in k:
            print(cfg, "options=%s" % os.environ.get("PROSPER_PT_DEBUG_OPTIONS"), k, " ".join("%s=%.3e" % (c, v.get(c, 0.0)) for c in counters), "us=%.0f" % v["us_per_launch"], flush=True)
