#!/bin/bash
# Builds kernel variants side by side: tools/build_variants.sh name1 "-DFLAGS" name2 "-DFLAGS" ...
# Each lands in raytracing_amd/lib/variants/<name>/librt_amd.so (select with RT_AMD_LIB).
set -e
cd "$(dirname "$0")/../raytracing_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  make -s LIBDIR=../lib/variants/$name BINDIR=../lib/variants/$name/bin EXTRA="$flags" > /dev/null
  echo built $name "($flags)"
done
