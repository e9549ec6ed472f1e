#!/bin/bash
# per-wave timeline of k_pt_persistent on one rank's share of the teapots frame (stamps build)
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export RADISH_HIP_LIB=$R/radish_pt_amd/csrc/variants/libradish_hip_stamps.so
for w in 1 2 4 8; do
  echo "== world $w" | tee -a $OUT/timeline.txt
  timeout -k 10 200 python3 scripts/wave_timeline.py teapots $w 0 >> $OUT/timeline.txt 2>&1
done
tail -60 $OUT/timeline.txt
