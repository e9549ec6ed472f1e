#!/bin/bash
# usage: scripts_pmc.sh <tag> <mode>   — separate PMC passes (rocprofv3 refuses mixing --pmc with tracing on this pool)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; MODE=$2
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $R
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --mode $MODE --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_LDS
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE
