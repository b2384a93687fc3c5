#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/profiles_r04/ (copy what is to be judged into profiles/):
#   r04_<config>_pmc.json                 rocprofv3 PMC per kernel, whole-batch launches each alone on the GPU (scripts/pmc_tools.py)
#   r04_<config>_kernel_stats.csv         rocprofv3 --kernel-trace --stats of scripts/quick_bench.py --single-chain (one chain, in
#                                         order: launches never overlap, so calls x average adds up to the steps' time)
#   r04_<config>_quick_bench.txt          the same command's own per-launch hipEvent times and step time
#   r04_bench.json / .err                 the default bench.py line (under 4 KB), r04_bench_detail.json the full result beside it
# usage: bash scripts/collect_profiles.sh [configs...]   (default: c2 c3 c4 helmet)
set -u
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_r04
mkdir -p "$OUT"
CONFIGS=${*:-c2 c3 c4 helmet}
export TMPDIR=/tmp
for c in $CONFIGS; do
  echo "== $c: PMC passes"; 
  python3 scripts/pmc_tools.py $c > "$OUT/r04_${c}_pmc.txt" 2>&1 && cp profiles/r04_${c}_pmc.json "$OUT/"
  echo "== $c: kernel trace"
  rm -rf /tmp/kt_$c
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_$c -o kt --output-format csv -- python3 "$ROOT/scripts/quick_bench.py" --config $c --single-chain --steps 20 > "$OUT/r04_${c}_quick_bench.txt" 2>&1)
  f=$(find /tmp/kt_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r04_${c}_kernel_stats.csv"
  tail -2 "$OUT/r04_${c}_quick_bench.txt"
done
echo "== bench.py"
python3 bench.py --steps 20 --warmup 5 --detail "$OUT/r04_bench_detail.json" > "$OUT/r04_bench.json" 2> "$OUT/r04_bench.err"
echo "bench rc=$?"
