#!/bin/bash
# Copies what scripts/collect_profiles.sh left in gpurun_out/profiles_r04/ into profiles/ (the tracked, judged set).
set -e
cd "$(dirname "$0")/.."
cp gpurun_out/profiles_r04/r04_*_pmc.json gpurun_out/profiles_r04/r04_*_kernel_stats.csv profiles/
cp gpurun_out/profiles_r04/r04_bench.json profiles/r04_bench.json
cp gpurun_out/profiles_r04/r04_bench_detail.json profiles/r04_bench_detail.json
for c in c2 c3 c4 helmet; do grep -v "^W2026\|^E2026" gpurun_out/profiles_r04/r04_${c}_quick_bench.txt | tail -2 > profiles/r04_${c}_single_chain_times.txt; done
(echo "# rocprofv3 PMC per kernel, round 4 final sources (scripts/pmc_tools.py c2 c3 c4 helmet; whole-batch launches, each alone on the GPU; 1920x1080 x 8 spp)"; for c in c2 c3 c4 helmet; do cat gpurun_out/profiles_r04/r04_${c}_pmc.txt; done) > profiles/r04_pmc_summary.txt
python3 - <<'PY'
import json, sys
sys.path.insert(0, "scripts")
import pmc_tools
d = json.loads(open("profiles/r04_bench.json").read().strip().splitlines()[-1])
print("C2 %.1f Mpaths/s, %.3f ms; frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for n, c in d["configs"].items():
    print(" ", n, "%.3f ms  %.1f Mpaths/s  cpu %.2f" % (c["ms_per_step"], c["Mpaths_per_s"], c.get("cpu_Mpaths_per_s", 0.0)))
print("kernel sources sha16 now %s, profiles %s" % (pmc_tools.kernel_source_hash(), json.load(open("profiles/r04_c2_pmc.json"))["kernel_source_sha16"]))
PY
