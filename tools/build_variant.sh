#!/bin/bash
# Builds another copy of the library for A/B measurements in one GPU session:
#     tools/build_variant.sh NAME [GIT_REF [FILE ...]] [-- EXTRA_HIPCC_FLAGS]
# copies pyspeedy_amd/csrc to build_variants/NAME/csrc, replaces FILE ... (paths relative to pyspeedy_amd/csrc; default: every
# tracked source) by their content at GIT_REF, builds there and leaves build_variants/lib_NAME.so.  Select it at run time with
# PYSPEEDY_AMD_LIB=build_variants/lib_NAME.so (pyspeedy_amd/_lib.py).  build_variants/ is git-ignored and travels with gpurun.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
REF=""; FILES=(); EXTRA=""
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; EXTRA="$*"; break; fi
  if [ -z "$REF" ]; then REF=$1; else FILES+=("$1"); fi
  shift
done
D=$ROOT/build_variants/$NAME
rm -rf "$D" && mkdir -p "$D/pyspeedy_amd" "$D/include"
cp -r "$ROOT/pyspeedy_amd/csrc" "$D/pyspeedy_amd/csrc"
cp "$ROOT"/include/*.h "$D/include/"
rm -f "$D"/pyspeedy_amd/csrc/*.o
if [ -n "$REF" ]; then
  if [ ${#FILES[@]} -eq 0 ]; then
    mapfile -t FILES < <(cd "$ROOT" && git ls-tree --name-only "$REF" pyspeedy_amd/csrc/ | sed 's|pyspeedy_amd/csrc/||')
    for h in $(cd "$ROOT" && git ls-tree --name-only "$REF" include/ | grep '\.h$'); do git -C "$ROOT" show "$REF:$h" > "$D/$h"; done
  fi
  for f in "${FILES[@]}"; do git -C "$ROOT" show "$REF:pyspeedy_amd/csrc/$f" > "$D/pyspeedy_amd/csrc/$f"; done
fi
if [ -n "$EXTRA" ]; then
  make -s -C "$D/pyspeedy_amd/csrc" -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on $EXTRA"
else
  make -s -C "$D/pyspeedy_amd/csrc" -j8
fi
cp "$D/pyspeedy_amd/libpyspeedy_amd.so" "$ROOT/build_variants/lib_$NAME.so"
rm -rf "$D"
echo "built build_variants/lib_$NAME.so"
