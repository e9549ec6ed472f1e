#!/bin/bash
# scratch runner (round 3): experiment — the G-buffer pass launched beside ReSTIR's shadow walk (RADISH_GBUFFER_INSIDE=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zr; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
for v in 0 1; do
  RADISH_GBUFFER_INSIDE=$v timeout -k 10 200 python3 bench.py --workload restir --steps 16 > $OUT/bench_restir_$v.json 2> $OUT/bench_restir_$v.err; say "   inside=$v rc=$?"
  python3 -c "import json;d=json.loads(open('$OUT/bench_restir_$v.json').read().strip().splitlines()[-1]);print('   inside=$v ms_per_step',d['ms_per_step'],d['value'])" | tee -a $OUT/progress.log
done
say "default bench sub-record 4 with inside=1 (parity sample)"
RADISH_GBUFFER_INSIDE=1 timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-pipelined --no-traversal-only > $OUT/bench_default_1.json 2> $OUT/bench_default_1.err; say "   rc=$?"
python3 -c "
import json;d=json.loads(open('$OUT/bench_default_1.json').read().strip().splitlines()[-1]);c=d['configs']['4']
print('   cfg4',c['ms_per_step'],c['ms_per_step_host_blocking'],c['parity_sample_ok'])" | tee -a $OUT/progress.log
say "kernel trace"; RADISH_GBUFFER_INSIDE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_restir -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 8 > $OUT/trace_restir.log 2>&1; say "   rc=$?"
say done
