#!/usr/bin/env python3
"""Static instruction census of a gfx950 .s file (hipcc -S --cuda-device-only): per kernel totals."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
name = None
ops = None
for line in lines:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, ops = m.group(1), collections.Counter()
        continue
    if name is None:
        continue
    t = line.strip()
    if t.startswith("s_endpgm"):
        if len(sys.argv) < 3 or sys.argv[2] in name:
            total = sum(ops.values())
            valu = sum(v for k, v in ops.items() if k.startswith("v_"))
            print("%-60s total %5d valu %5d div %3d sqrt %3d rcp %3d setreg %3d vmem %3d lds %3d waitcnt %3d branch %3d" % (
                name[:60], total, valu, ops["v_div_fixup_f32"], ops["v_sqrt_f32"], ops["v_rcp_f32"],
                ops["s_setreg_imm32_b32"] + ops["s_setreg_b32"],
                sum(v for k, v in ops.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_"))),
                sum(v for k, v in ops.items() if k.startswith("ds_")), ops["s_waitcnt"],
                sum(v for k, v in ops.items() if k.startswith(("s_cbranch", "s_branch")))))
        name = None
        continue
    if not t or t.startswith((".", ";", "/")) or t.endswith(":"):
        continue
    ops[t.split()[0]] += 1
