#!/bin/bash
# Builder knob sweep (leaf size, SAH traversal cost) on C4 / FlightHelmet / C3, three frames in flight: does trading node
# visits for triangle tests pay?  Tooling (profiles/r03_leaf_sweep.txt); run on the GPU box from the repository root.
mkdir -p gpurun_out
out=gpurun_out/r03_leaf_sweep.txt
: > $out
# (the library reads PROSPER_PT_DEBUG_OPTIONS at prosper_pt_create only while PROSPER_PT_DEBUG=1: prosper_pt.h, debug options)
for v in "" "leafSize=2" "leafSize=1" "sahTraversalCost=0.5" "sahTraversalCost=2" "leafSize=2,sahTraversalCost=0.5"; do
  echo "== ${v:-default}" >> $out
  env PROSPER_PT_DEBUG=1 PROSPER_PT_DEBUG_OPTIONS="$v" python scripts/pipelined_bench.py c4 helmet c3 2>&1 | grep pipelined >> $out || exit 1
done
