#!/bin/bash
# scratch runner (round 3): scalar-cache counters of the packet kernels, then the GPU suite on the final tree
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zx; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
for pass in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQC_DCACHE_INPUT_VALID_READYB" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  n=$(echo $pass | cut -d' ' -f1); say "pass $n"
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc/$n -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 4 > $OUT/pmc_$n.log 2>&1 || say "   pass $n failed: $(tail -2 $OUT/pmc_$n.log)"
done
python3 scripts/pmc_per_frame.py $OUT/pmc 4 > $OUT/pmc_summary.txt 2>&1
say "[gpu tests]"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; say "   rc=$? $(tail -1 $OUT/gpu_tests.log)"
say done
