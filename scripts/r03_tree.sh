#!/bin/bash
# Experiment r03_g: shared-tree walks in k_wf_trace.  usage: r03_tree.sh <tag>
TAG=$1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[0] config tests (teapots: tree by default)"; timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config3 or config1" > $OUT/tests.log 2>&1; say "   rc=$? $(tail -1 $OUT/tests.log)"
b() { # label env lib mode scene
  local L=$1 T=$2 LIB=$3 M=$4 S=$5
  R1=$(RADISH_TREE=$T RADISH_HIP_LIB=$LIB timeout -k 10 200 python3 bench.py --mode $M --scene $S --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  say "$L $S $M: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"])' 2>/dev/null || echo FAILED)"
}
V=$R/radish_pt_amd/csrc/variants/libradish_hip_wt1.so
for S in teapots cornell; do
  for M in wavefront_sort2 wavefront2 wavefront_sort; do
    b threaded 0 "" $M $S
    b tree7 1 "" $M $S
    b tree6 1 $V $M $S
  done
done
say done
