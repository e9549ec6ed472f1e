#!/bin/bash
# Experiment r03_l: sibling-pair walks in k_pt_persistent.  usage: r03_persist.sh <tag>
TAG=$1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[0] persistent-mode tests"; timeout -k 10 400 python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "config5 or config3 or path_trace" > $OUT/tests.log 2>&1; say "   rc=$? $(tail -1 $OUT/tests.log)"
b() { # label env lib scene W H
  local L=$1 T=$2 LIB=$3 S=$4 W=$5 H=$6
  R1=$(RADISH_PAIRS=$T RADISH_HIP_LIB=$LIB timeout -k 10 120 python3 bench.py --mode persistent --scene $S --width $W --height $H --steps 6 --warmup 3 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  say "$L $S ${W}x$H persistent: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"])' 2>/dev/null || echo FAILED)"
}
V=$R/radish_pt_amd/csrc/variants
for sc in "teapots 1920 1080" "teasets_1m 3840 2160" "cornell 1920 1080"; do
  b threaded 0 "" $sc
  b pairs3 1 "" $sc
  b pairs2 1 $V/libradish_hip_pp2.so $sc
done
say done
