#!/bin/bash
TAG=$1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
b() { # label env lib mode scene
  local L=$1 T=$2 LIB=$3 M=$4 S=$5
  R1=$(RADISH_TREE=$T RADISH_HIP_LIB=$LIB timeout -k 10 200 python3 bench.py --mode $M --scene $S --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  say "$L $S $M: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"], "parity", d.get("parity_check"))' 2>/dev/null || echo FAILED)"
}
V=$R/radish_pt_amd/csrc/variants
for M in wavefront_sort2 wavefront_sort; do
  b threaded 0 "" $M teapots
  for v in wt1 wt1_l16 wt1_l12 wt1_l8 wt7_l12; do b $v 1 $V/libradish_hip_$v.so $M teapots; done
done
say done
