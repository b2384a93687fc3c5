#!/usr/bin/env python3
"""PMC comparison of debug-option settings (tooling): the three passes of scripts/pmc_tools.py per setting, through the
library's environment gate (PROSPER_PT_DEBUG=1 PROSPER_PT_DEBUG_OPTIONS=...), each launch alone on the GPU.

    python scripts/pmc_options_ab.py <config> "<options>" ["<options>" ...]       ("" = the defaults)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_tools  # noqa: E402

config = sys.argv[1]
for options in sys.argv[2:]:
    os.environ["PROSPER_PT_DEBUG"] = "1"
    os.environ["PROSPER_PT_DEBUG_OPTIONS"] = options
    out = pmc_tools.collect(config, timeout=300)
    for k, v in sorted(out["kernels"].items()):
        print("%s [%s] %-20s VALU %.3e  lane_util %.2f  cyc/inst/simd %.2f  cu_busy %.2f  FETCH %.3f GB  HBM %.3f GB  %.0f us" % (
            config, options or "defaults", k, v["valu_insts_per_launch"], v["lane_util"], v["cycles_per_valu_inst_per_simd_profiled"],
            v["cu_busy"], v["FETCH_SIZE_KB_per_launch"] * 1024 / 1e9, v["hbm_bytes_per_launch"] / 1e9, v["us_per_launch_profiled"]), flush=True)
