#!/bin/bash
# scratch runner (round 3): shared normalize / length / pdf terms in the light-sampling code — parity, then config 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03ze; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[0] gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?; say "   rc=$rc $(tail -1 $OUT/gpu_tests.log)"
[ $rc -ne 0 ] && { tail -30 $OUT/gpu_tests.log; exit 1; }
say "[1] restir workload"; timeout -k 10 200 python3 bench.py --workload restir --steps 16 > $OUT/bench_restir.json 2> $OUT/bench_restir.err; say "   rc=$?"
python3 -c "import json;d=json.loads(open('$OUT/bench_restir.json').read().strip().splitlines()[-1]);print('   ms_per_step',d['ms_per_step'],d['value'])" | tee -a $OUT/progress.log
say "[2] kernel trace restir"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_restir -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 8 > $OUT/trace_restir.log 2>&1; say "   rc=$?"
say "[3] default bench"; timeout -k 10 400 python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; say "   rc=$?"
python3 -c "
import json;d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]);c=d['configs']
print('   headline',d['ms_per_step'],d['parity_check'] and d['parity_check']['bit_exact'],'cfg2',c['2']['ms_per_step'],'cfg4',c['4']['ms_per_step'],c['4'].get('ms_per_step_host_blocking'),c['4']['ms_gbuffer_kernels'],c['4']['ms_restir_kernels'],'pipelined',d['pipelined']['ms_per_step'])" | tee -a $OUT/progress.log
say done
