#!/bin/bash
# Experiment r03_i: sibling-pair walks in k_wf_trace.  usage: r03_tree2.sh <tag>
TAG=$1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[0] config tests (teapots: pairs by default)"; timeout -k 10 240 python3 -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config3 or config1" > $OUT/tests.log 2>&1; say "   rc=$? $(tail -1 $OUT/tests.log)"
b() { # label env lib mode scene
  local L=$1 T=$2 LIB=$3 M=$4 S=$5
  R1=$(RADISH_PAIRS=$T RADISH_HIP_LIB=$LIB timeout -k 10 90 python3 bench.py --mode $M --scene $S --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  say "$L $S $M: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"])' 2>/dev/null || echo FAILED)"
}
V=$R/radish_pt_amd/csrc/variants
for M in wavefront_sort2 wavefront2 wavefront_sort; do
  b threaded 0 "" $M teapots
  b pairs 1 "" $M teapots
  for v in pl4 pl16; do b $v 1 $V/libradish_hip_$v.so $M teapots; done
done
for M in wavefront_sort2 wavefront_sort; do b threaded 0 "" $M cornell; b pairs 1 "" $M cornell; done
say done
