#!/bin/bash
# scratch runner (round 3): final check of the tree — GPU suite, restir workload, default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zu; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[0] gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?; say "   rc=$rc $(tail -1 $OUT/gpu_tests.log)"
[ $rc -ne 0 ] && { tail -40 $OUT/gpu_tests.log; exit 1; }
say "[1] smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $OUT/progress.log
say "[2] restir workload"; timeout -k 10 200 python3 bench.py --workload restir --steps 16 > $OUT/bench_restir.json 2> $OUT/bench_restir.err; say "   rc=$?"
python3 -c "import json;d=json.loads(open('$OUT/bench_restir.json').read().strip().splitlines()[-1]);print('   ms_per_step',d['ms_per_step'],d['value'])" | tee -a $OUT/progress.log
say "[3] default bench"; timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; say "   rc=$?"
python3 -c "
import json;d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]);c=d['configs']
print('   headline',d['ms_per_step'],d['value'],d['roofline']['frac'],d['parity_check']['bit_exact'],'pipelined',d['pipelined']['ms_per_step'],'cfg2',c['2']['ms_per_step'],'cfg4',c['4']['ms_per_step'],c['4']['ms_per_step_host_blocking'],'cfg5',c['5_scene_one_gpu']['ms_per_step'])" | tee -a $OUT/progress.log
say done
