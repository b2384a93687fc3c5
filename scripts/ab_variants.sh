#!/bin/bash
# A/B of library variants (scripts/build_variant.sh) on the GPU box: frames in flight, 1920x1080 x 8 spp.  Tooling.
#   scripts/ab_variants.sh <out file> "<configs>" <variant> [<variant> ...]        ("default" = prosper_amd/libprosper_pt.so)
out=$1; configs=$2; shift 2
mkdir -p "$(dirname "$out")"
: > "$out"
for v in "$@"; do
  if [ "$v" = default ]; then lib=prosper_amd/libprosper_pt.so; else lib=build/variants/lib_$v.so; fi
  echo "== $v" >> "$out"
  LIB=$lib python scripts/pipelined_bench.py $configs 2>&1 | grep "ranks 1 " >> "$out" || exit 1
done
