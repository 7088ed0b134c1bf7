#!/bin/bash
# Builds a variant of libpyspeedy_amd.so with extra compiler flags into build_variants/lib_<name>.so (not committed; travels
# to the GPU box).  Used for A/B measurements in one gpurun session: PYSPEEDY_AMD_LIB=build_variants/lib_<name>.so python bench.py
#     tools/build_variant.sh <name> [extra hipcc flags, e.g. -DSPD_EXP=3] [-- file.hip ...  (only these get the flags)]
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/pyspeedy_amd/csrc
OBJ=$ROOT/build_variants/obj_$NAME
mkdir -p $OBJ
FLAGS=()
ONLY=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; ONLY=("$@"); break; fi
  FLAGS+=("$1"); shift
done
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on"
pids=()
for f in capi transforms specops physics dynamics model surface sppt stream_apart driver_backend_hip; do
  extra="${FLAGS[*]}"
  if [ ${#ONLY[@]} -gt 0 ]; then
    extra=""
    for o in "${ONLY[@]}"; do [ "$o" == "$f.hip" ] && extra="${FLAGS[*]}"; done
    # unchanged objects are reused from the main build
    if [ -z "$extra" ] && [ -f $SRC/$f.o ]; then cp $SRC/$f.o $OBJ/$f.o; continue; fi
  fi
  /opt/rocm/bin/hipcc $BASE $extra -c $SRC/$f.hip -o $OBJ/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
cp $SRC/tables.o $SRC/surface_host.o $SRC/driver.o $OBJ/   # (host-only C++: always the main build's objects)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/build_variants/lib_$NAME.so $OBJ/*.o
rm -rf $OBJ
echo "built build_variants/lib_$NAME.so"
