#!/bin/bash
# scratch runner (round 3): wgTraceWholeSliced (two windows per round trip in the workgroup-per-ray launches) — parity, then timings
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zs; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[0] gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?; say "   rc=$rc $(tail -1 $OUT/gpu_tests.log)"
[ $rc -ne 0 ] && { tail -40 $OUT/gpu_tests.log; exit 1; }
say "[1] restir workload"; timeout -k 10 200 python3 bench.py --workload restir --steps 16 > $OUT/bench_restir.json 2> $OUT/bench_restir.err; say "   rc=$?"
python3 -c "import json;d=json.loads(open('$OUT/bench_restir.json').read().strip().splitlines()[-1]);print('   ms_per_step',d['ms_per_step'],d['value'])" | tee -a $OUT/progress.log
say "[2] kernel trace restir"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_restir -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 8 > $OUT/trace_restir.log 2>&1; say "   rc=$?"
python3 - <<'P' | tee -a $OUT/progress.log
import csv,glob
f=glob.glob('/root/repo/gpurun_out/r03zs/trace_restir/runc/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']; n=n[:n.find('(')][-40:]
    if 'packet' in n or 'literal' in n or 'wg_list' in n or 'walk_pair' in n: print(f"   {n:40s} avg {float(r['AverageNs'])/1e3:7.1f} min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
P
say "[3] rank shares restir (config 5 scene 4K + config 4)"; timeout -k 10 600 python3 scripts/partition_times_restir.py > $OUT/partition_times_restir.txt 2>&1; grep '"tile": 128' $OUT/partition_times_restir.txt | grep -v rows | tee -a $OUT/progress.log
say done
