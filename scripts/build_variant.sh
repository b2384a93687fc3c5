#!/bin/bash
# Builds build/variants/lib_<name>.so: the library with extra device-code defines (A/B experiments; tooling).
#   scripts/build_variant.sh <name> "<extra hipcc flags>"      e.g.  scripts/build_variant.sh wpe4 "-DPPT_TRACE_WPE(s)=4"
#   HOST_DEFS="-DPPT_SAH_BINS=32" scripts/build_variant.sh bins32 ""     also recompiles the hierarchy builder with these defines
set -e
cd "$(dirname "$0")/../prosper_amd/csrc"
name=$1; shift
mkdir -p ../../build/variants /tmp/variant_$name
make -s ../libprosper_pt.so >/dev/null
for f in pt_kernels pt_wavefront; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -fno-slp-vectorize --offload-arch=gfx950 "$@" -c $f.hip -o /tmp/variant_$name/$f.o &
done
bvh=bvh_build.o
if [ -n "$HOST_DEFS" ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $HOST_DEFS -c bvh_build.cpp -o /tmp/variant_$name/bvh_build.o &
  bvh=/tmp/variant_$name/bvh_build.o
fi
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/variants/lib_$name.so prosper_pt.o pt_geometry.o pt_materials.o pt_tiling.o $bvh host/camera.o host/rt_reference.o host/tiled_rt_reference.o host/tone_map.o /tmp/variant_$name/pt_kernels.o /tmp/variant_$name/pt_wavefront.o -ldl
echo built build/variants/lib_$name.so
