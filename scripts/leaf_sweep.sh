#!/bin/bash
# Builder knob sweep (leaf size, SAH traversal cost) on C4 / FlightHelmet / C3, three frames in flight: does trading node
# visits for triangle tests pay?  Tooling (profiles/r03_leaf_sweep.txt); run on the GPU box from the repository root.
mkdir -p gpurun_out
out=gpurun_out/r03_leaf_sweep.txt
: > $out
for v in "" "PROSPER_PT_DEBUG_LEAF=2" "PROSPER_PT_DEBUG_LEAF=1" "PROSPER_PT_DEBUG_SAH_TC=0.5" "PROSPER_PT_DEBUG_SAH_TC=2" "PROSPER_PT_DEBUG_LEAF=2 PROSPER_PT_DEBUG_SAH_TC=0.5"; do
  echo "== ${v:-default}" >> $out
  env $v python scripts/pipelined_bench.py c4 helmet c3 2>&1 | grep pipelined >> $out || exit 1
done
