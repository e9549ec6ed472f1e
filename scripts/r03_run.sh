#!/bin/bash
# scratch runner (round 3): packet walks for primary rays (G-buffer, ReSTIR walk 1, wavefront raygen) + high-priority literal stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zh; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[0] gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?; say "   rc=$rc $(tail -1 $OUT/gpu_tests.log)"
[ $rc -ne 0 ] && { tail -40 $OUT/gpu_tests.log; exit 1; }
say "[1] restir workload: packets+priority / packets, no priority / neither"
for v in "1 1" "1 0"; do set -- $v; RADISH_PACKETS=$1 RADISH_LIT_PRIORITY=$2 timeout -k 10 200 python3 bench.py --workload restir --steps 16 > $OUT/bench_restir_p$1$2.json 2> $OUT/bench_restir_p$1$2.err; say "   rc=$?"
python3 -c "import json;d=json.loads(open('$OUT/bench_restir_p$1$2.json').read().strip().splitlines()[-1]);print('   packets=$1 prio=$2 ms_per_step',d['ms_per_step'],d['value'])" | tee -a $OUT/progress.log; done
say "[2] kernel trace restir"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_restir -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 8 > $OUT/trace_restir.log 2>&1; say "   rc=$?"
say "[3] default bench, packets on / off"
for p in 1 0; do RADISH_PACKETS=$p timeout -k 10 400 python3 bench.py --no-cpu-baseline > $OUT/bench_default_p$p.json 2> $OUT/bench_default_p$p.err; say "   rc=$?"
python3 -c "
import json;d=json.loads(open('$OUT/bench_default_p$p.json').read().strip().splitlines()[-1]);c=d['configs']
print('   packets=$p headline',d['ms_per_step'],d['roofline']['frac'],'pipelined',d['pipelined']['ms_per_step'],'cfg2',c['2']['ms_per_step'],'cfg4',c['4']['ms_per_step'],c['4'].get('ms_per_step_host_blocking'),c['4']['ms_gbuffer_kernels'],c['4']['ms_restir_kernels'],c['4']['parity_sample_ok'],'cfg5',c['5_scene_one_gpu']['ms_per_step'])" | tee -a $OUT/progress.log; done
say "[4] kernel trace teapots frames"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_frames -- python3 scripts/pmc_frames.py teapots wavefront_sort2 1920 1080 6 > $OUT/trace_frames.log 2>&1; say "   rc=$?"
say done
