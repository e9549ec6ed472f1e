#!/bin/bash
# scratch runner (round 3): pathTrace's primary rays as packets in a launch of their own (k_primary_packet) — parity, then A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zo; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[0] gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; rc=$?; say "   rc=$rc $(tail -1 $OUT/gpu_tests.log)"
[ $rc -ne 0 ] && { tail -40 $OUT/gpu_tests.log; exit 1; }
say "[1] default bench, primary packets on / off"
for p in 1 0; do RADISH_PRIMARY_PACKETS=$p timeout -k 10 400 python3 bench.py --no-cpu-baseline > $OUT/bench_default_p$p.json 2> $OUT/bench_default_p$p.err; say "   rc=$?"
python3 -c "
import json;d=json.loads(open('$OUT/bench_default_p$p.json').read().strip().splitlines()[-1]);c=d['configs']
print('   primary=$p headline',d['ms_per_step'],d['roofline']['frac'],d['parity_check'],'pipelined',d['pipelined']['ms_per_step'],'cfg2',c['2']['ms_per_step'],'cfg4',c['4']['ms_per_step'],'cfg5',c['5_scene_one_gpu']['ms_per_step'])" | tee -a $OUT/progress.log; done
say "[2] other structures"; for m in wavefront2 wavefront_sort persistent; do for p in 1 0; do RADISH_PRIMARY_PACKETS=$p timeout -k 10 200 python3 bench.py --mode $m --steps 10 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only > $OUT/b_${m}_$p.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$OUT/b_${m}_$p.json').read().strip().splitlines()[-1]);print('   $m primary=$p',d['ms_per_step'])" | tee -a $OUT/progress.log; done; done
say "[2b] rank shares, persistent"; for p in 1 0; do RADISH_PRIMARY_PACKETS=$p timeout -k 10 300 python3 scripts/partition_times.py teapots 1920 1080 persistent > $OUT/share_p$p.txt 2>&1; grep '"mode"' $OUT/share_p$p.txt | grep -v rows | sed "s/^/   primary=$p /" | tee -a $OUT/progress.log; done
say "[3] kernel trace teapots frames"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_frames -- python3 scripts/pmc_frames.py teapots wavefront_sort2 1920 1080 6 > $OUT/trace_frames.log 2>&1; say "   rc=$?"
say done
