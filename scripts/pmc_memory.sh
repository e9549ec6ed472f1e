#!/bin/bash
# memory-pipeline counters for one bench mode:  scripts_pmc2.sh <tag> <mode>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; MODE=$2; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd $R
run() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --mode $MODE --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run m1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
run m2 TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum
run m3 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run m4 TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TA_TOTAL_WAVEFRONTS_sum
run m5 TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum
run m6 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run m7 GRBM_GUI_ACTIVE TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum
