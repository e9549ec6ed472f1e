#!/bin/bash
# scratch runner (round 3): packet walk with a visit budget, lanes finishing on their own — parity, then budgets 96 / 128 / 192 / 256 / 384
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zw; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[0] gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?; say "   rc=$rc $(tail -1 $OUT/gpu_tests.log)"
[ $rc -ne 0 ] && { tail -40 $OUT/gpu_tests.log; exit 1; }
for b in 192 96 128 256 384; do
  lib=$R/radish_pt_amd/csrc/variants/libradish_hip_pb$b.so; [ $b = 192 ] && lib=$R/radish_pt_amd/csrc/libradish_hip.so
  RADISH_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$b -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 8 > $OUT/trace_$b.log 2>&1
  python3 - $b <<'P' | tee -a $OUT/progress.log
import csv,glob,sys
g=sys.argv[1]
f=glob.glob(f'/root/repo/gpurun_out/r03zw/trace_{g}/runc/*_kernel_stats.csv')[0]
o=[]
for r in csv.DictReader(open(f)):
    n=r['Name']; n=n[:n.find('(')][-28:]
    if 'packet' in n: o.append(f"{n.strip()} avg {float(r['AverageNs'])/1e3:6.1f} min {float(r['MinNs'])/1e3:6.1f}")
print(f"   budget={g}: "+" | ".join(o))
P
  RADISH_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --workload restir --steps 16 > $OUT/bench_restir_$b.json 2>/dev/null
  python3 -c "import json;d=json.loads(open('$OUT/bench_restir_$b.json').read().strip().splitlines()[-1]);print('   budget=$b restir ms_per_step',d['ms_per_step'])" | tee -a $OUT/progress.log
done
say done
