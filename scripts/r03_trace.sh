#!/bin/bash
# kernel traces of the teapots frame, pairs vs threaded, one pipeline and three.  usage: r03_trace.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for P in 1 0; do for M in wavefront_sort wavefront_sort2; do
  export RADISH_PAIRS=$P
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_${M}_p$P -- python3 scripts/pmc_frames.py teapots $M 1920 1080 6 > $OUT/t_${M}_p$P.log 2>&1
  echo "== $M pairs=$P" >> $OUT/summary.txt
  f=$(find $OUT/t_${M}_p$P -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-200 >> $OUT/summary.txt
done; done
cat $OUT/summary.txt
