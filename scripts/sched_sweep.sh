#!/bin/bash
# Re-sweep of the stream scheduler's thresholds after round 3's changes (pt_trace_stream.hpp PPT_TRI_BEATS_NODE, PPT_REFILL,
# PPT_CONTINUE), three frames in flight.  Tooling (profiles/r03_trace_experiments.txt); build the variants first:
#   scripts/build_variant.sh tri10 "-DPPT_TRI_BEATS_NODE(t,n)=((t)>(n))"      tri20: ((t)>2u*(n))
#   scripts/build_variant.sh ref24 "-DPPT_REFILL(r,best)=((r)>=24u)"           ref40 / ref48 likewise
#   scripts/build_variant.sh con2  "-DPPT_CONTINUE(c,c0,other)=(2u*(c)>(c0))"  con4 / con5 likewise
mkdir -p gpurun_out
out=gpurun_out/r03_sched_sweep.txt
: > $out
for v in default tri10 tri20 ref24 ref40 ref48 con2 con4 con5 default; do
  echo "== $v" >> $out
  if [ $v = default ]; then lib=prosper_amd/libprosper_pt.so; else lib=build/variants/lib_$v.so; fi
  LIB=$lib python scripts/pipelined_bench.py c2 c3 c4 helmet 2>&1 | grep "ranks 1 pipelined" >> $out || exit 1
done
