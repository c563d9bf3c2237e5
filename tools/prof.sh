#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> <prof_run.py args...>
# Runs a kernel-trace/stats pass and separate PMC passes; results under gpurun_out/prof_<tag>/.
set -e
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
run="python3 $OLDPWD/tools/prof_run.py $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $run > $out/stats.log 2>&1
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
            "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
            "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_READ_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $out/pmc$i -- $run > $out/pmc$i.log 2>&1
done
echo done
