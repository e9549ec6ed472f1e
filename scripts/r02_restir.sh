#!/bin/bash
# ReSTIR config 4: split vs fused pass 1, with a rocprofv3 per-kernel table of the split frame.  usage: r02_restir.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
echo "[1] tests"; timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_collectives.py -x -q -m gpu -k "restir or gbuffer or mirror or degenerate" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log
grep -q "rc=0" $OUT/tests.log || exit 1
echo "[2] split"; timeout -k 10 300 python3 scripts/bench_restir.py 2>&1 | tail -2 > $OUT/restir_split.json; cat $OUT/restir_split.json | cut -c1-420
echo "[3] fused"; RADISH_RESTIR_FUSED=1 timeout -k 10 300 python3 scripts/bench_restir.py 2>&1 | tail -2 > $OUT/restir_fused.json; cat $OUT/restir_fused.json | cut -c1-420
echo "[4] rocprof"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_restir -- python3 scripts/bench_restir.py > $OUT/trace_restir.log 2>&1
f=$(find $OUT/trace_restir -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-160
