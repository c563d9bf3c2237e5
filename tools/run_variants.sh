#!/bin/bash
# On the GPU box: tools/run_variants.sh <prof_run args> -- name1 name2 ...   (one line per variant)
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
for v in "$@"; do
  out=$(RT_AMD_LIB=$PWD/raytracing_amd/lib/variants/$v/librt_amd.so timeout -k 10 200 python tools/prof_run.py "${args[@]}" 2>&1 | grep "ms/launch" | cut -c1-150) || exit 1
  echo "$v: $out"
done
