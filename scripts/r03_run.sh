#!/bin/bash
# scratch runner (round 3): threaded fallback validation, literal-stream priority experiment, restir timing protocol
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03za; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[1] restir workload, default and RADISH_LIT_PRIORITY=1"
for p in 0 1; do
  RADISH_LIT_PRIORITY=$p timeout -k 10 200 python3 bench.py --workload restir --steps 16 > $OUT/restir_prio$p.json 2> $OUT/restir_prio$p.err; say "   prio=$p rc=$?"
  python3 -c "import json;d=json.loads(open('$OUT/restir_prio$p.json').read().strip().splitlines()[-1]);print('   ms_per_step',d['ms_per_step'],d['value'])" | tee -a $OUT/progress.log
done
say "[2] G-buffer + restir kernel trace with priority"
RADISH_LIT_PRIORITY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_restir_prio -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 8 > $OUT/trace_restir_prio.log 2>&1; say "   rc=$?"
say "[3] default bench with priority (wavefront path is not touched; config 4/5 sub-records are)"
RADISH_LIT_PRIORITY=1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-pipelined > $OUT/bench_prio1.json 2> $OUT/bench_prio1.err; say "   rc=$?"
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-pipelined > $OUT/bench_prio0.json 2> $OUT/bench_prio0.err; say "   rc=$?"
for p in 0 1; do python3 -c "
import json;d=json.loads(open('$OUT/bench_prio$p.json').read().strip().splitlines()[-1]);c=d['configs']
print('   prio $p: headline',d['ms_per_step'],'cfg2',c['2']['ms_per_step'],'cfg4',c['4']['ms_per_step'],c['4'].get('ms_per_step_host_blocking'),c['4']['ms_gbuffer_kernels'],c['4']['ms_restir_kernels'],'cfg5',c['5']['ms_per_step'])" | tee -a $OUT/progress.log; done
say "[4] threaded fallback: GPU suite with RADISH_PAIRS=0"
RADISH_PAIRS=0 timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests_threaded.log 2>&1; say "   rc=$? $(tail -1 $OUT/gpu_tests_threaded.log)"
say done
