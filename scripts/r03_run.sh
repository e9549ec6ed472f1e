#!/bin/bash
# scratch runner (round 3): PMC on config 4's frame with the packet kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zk; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"; do
  n=$(echo $pass | cut -d' ' -f1); say "pass $n"
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc/$n -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 4 > $OUT/pmc_$n.log 2>&1 || say "   pass $n failed"
done
python3 scripts/pmc_per_frame.py $OUT/pmc 4 > $OUT/pmc_restir_summary.txt 2>&1
say done
