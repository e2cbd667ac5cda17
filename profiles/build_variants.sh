#!/bin/bash
# Build libcloudsc2_hip variants with different -D switches into build/variants/ (git-ignored, but shipped to the GPU box) (dev tool).
#   bash profiles/build_variants.sh name1 "-DCS2_NL_FEXP=0" name2 "-DCS2_NL_PINX=0" ...
set -e
SRC=gt4py_dwarf_p_cloudsc2_tl_ad_amd/csrc
OUT=build/variants
mkdir -p $OUT
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  d=$(mktemp -d)
  for f in capi nl tl ad aux; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c $SRC/cloudsc2_$f.hip -o $d/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o $OUT/lib_$name.so
  rm -rf $d
  echo built $OUT/lib_$name.so
done
