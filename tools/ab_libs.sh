#!/bin/bash
# On the GPU box: the side workloads with two builds of the library, alternating, in one call (boxes differ by up
# to 10 %, so only runs of one call compare):  tools/ab_libs.sh <variant dir name under raytracing_amd/lib/variants> [workloads...]
old=$1; shift
for w in "$@"; do
  for rep in 1 2; do
    for which in new old; do
      if [ $which = old ]; then export RT_AMD_LIB=$PWD/raytracing_amd/lib/variants/$old/librt_amd.so RT_AMD_LIB_OLD=1; else unset RT_AMD_LIB RT_AMD_LIB_OLD; fi
      line=$(timeout -k 10 300 python3 bench.py --workload $w --steps 2 --no-cpu-baseline --no-primary 2>/dev/null) || exit 1
      echo "$w $which $(echo "$line" | python3 -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])")"
    done
  done
done
