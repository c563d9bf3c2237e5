#!/bin/bash
# Profiles the bench command on the GPU box (run from the repo root):
#   bash tools/profile_bench.sh <round-tag> [extra bench.py args]
# Pass 1: rocprofv3 --kernel-trace --stats of `python3 bench.py` (the command the driver runs).
# Pass 2..: separate --pmc passes (FETCH_SIZE / WRITE_SIZE / cache and SQ counters) of a shorter run of the same
# workload: a warm-up step and a timed step of 512 spp each = eight wavefront passes of 128 frames of samples (the
# same launch sizes as the full run); the summary takes the first pass (nothing sorted yet) and the last one (the
# sort tuner has settled: the state bench.py's timed steps run in).
# Raw output goes to gpurun_out/profile_<tag>/; tools/summarize_profile.py turns it into profiles/*.
set -e
tag=$1; shift
root=$PWD
out=$root/gpurun_out/profile_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/stats.log 2>&1
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -s KILL 600 rocprofv3 --pmc $ctrs --output-format csv -d $out/pmc$i -- python3 $root/bench.py --steps 1 --warmup 1 --spp 512 --no-cpu-baseline --no-primary --no-counting "$@" > $out/pmc$i.log 2>&1
done
echo profile done
