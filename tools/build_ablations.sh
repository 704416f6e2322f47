#!/bin/bash
# Development aid: builds probe variants of libgpsgs_hip.so (GSR_ABL_* defines; never part of a product build) next to the real one, as
# gps-gaussian_amd/lib/abl/libgpsgs_hip_<name>.so (git-ignored; they travel to the GPU box with the snapshot).  tools/stage_times.py --lib <path> loads one.
set -e
cd "$(dirname "$0")/../gps-gaussian_amd/csrc"
mkdir -p ../lib/abl
for v in "$@"; do
  make -s clean OUT=../lib/abl/$v >/dev/null 2>&1 || true
  make -s -j8 OUT=../lib/abl/$v HIPFLAGS_EXTRA="-D$v" all
  cp ../lib/abl/$v/libgpsgs_hip.so ../lib/abl/libgpsgs_hip_$v.so
  rm -rf ../lib/abl/$v
  echo "built lib/abl/libgpsgs_hip_$v.so"
done
