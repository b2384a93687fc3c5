#!/bin/bash
# A/B of debug options on the GPU box (PROSPER_PT_DEBUG=1 PROSPER_PT_DEBUG_OPTIONS=...): frames in flight, 1920x1080 x 8 spp.
#   scripts/ab_options.sh <out file> "<configs>" "<options or empty>" ["<options>" ...]
out=$1; configs=$2; shift 2
mkdir -p "$(dirname "$out")"
: > "$out"
for v in "$@"; do
  echo "== ${v:-default}" >> "$out"
  PROSPER_PT_DEBUG=1 PROSPER_PT_DEBUG_OPTIONS="$v" python scripts/pipelined_bench.py $configs 2>&1 | grep "ranks 1 " >> "$out" || exit 1
done
