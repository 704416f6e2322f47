#!/bin/bash
# Development aid: lib/abl/libgpsgs_hip_<name>.so = the product library with gsr_preprocess / gsr_binning / gsr_composite_tiles rebuilt under extra -D flags
# (the other objects are reused from lib/obj).  tools/stage_times.py --lib <path> and GPSGS_LIB=<path> load one.  Usage: tools/build_variant.sh <name> "<flags>"
set -e
name=$1; flags=$2
cd "$(dirname "$0")/../gps-gaussian_amd/csrc"
make -s -j8 all
out=../lib/abl/obj_$name; mkdir -p $out
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off -fno-slp-vectorize $flags -c gsr_preprocess.hip -o $out/gsr_preprocess.o &
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off -mllvm -simplifycfg-sink-common=false $flags -c gsr_binning.hip -o $out/gsr_binning.o &
/opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize $flags -c gsr_composite_tiles.hip -o $out/gsr_composite_tiles.o &
/opt/rocm/bin/hipcc $COMMON -fno-slp-vectorize $flags -c gsr_composite.hip -o $out/gsr_composite.o &
/opt/rocm/bin/hipcc $COMMON $flags -c capi.hip -o $out/capi.o &
wait
objs=""
for o in ../lib/obj/*.o; do b=$(basename $o); if [ -f $out/$b ]; then objs="$objs $out/$b"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../lib/abl/libgpsgs_hip_$name.so
rm -rf $out
echo "built lib/abl/libgpsgs_hip_$name.so"
