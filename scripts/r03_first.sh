#!/bin/bash
# round 3, first GPU call: the GPU suite, the new default bench line, per-rank partition times.  usage: r03_first.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[1] gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; say "   rc=$? $(tail -1 $OUT/gpu_tests.log)"
say "[2] default bench"; timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; say "   rc=$?"; cut -c1-400 $OUT/bench_default.json
say "[3] partition times"; timeout -k 10 300 python3 scripts/partition_times.py > $OUT/partition_times.txt 2>&1; tail -1 $OUT/partition_times.txt | cut -c1-600
say done
