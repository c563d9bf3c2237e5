#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/prof_lat.sh <tag> <prof_run.py args...>
# Issue/latency counters of the traversal kernels: three PMC passes, results under gpurun_out/lat_<tag>/.
set -e
tag=$1; shift
out=$PWD/gpurun_out/lat_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
run="python3 $OLDPWD/tools/prof_run.py $*"
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH" \
            "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d $out/pmc$i -- $run > $out/pmc$i.log 2>&1
done
python3 - $out <<'PY'
import collections, csv, glob, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for s in ("wf_extend", "wf_shade", "k_render_mega", "k_primary"):
            if s in k:
                per[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in per:
    for c in sorted(per[k]):
        v = per[k][c]
        print("%-14s %-34s %14.6g  (%d launches)" % (k, c, sum(v) / len(v), len(v)))
PY
