#!/bin/bash
# Config 3's structure: wavefront pipeline, unsorted / sorted, one pipeline / two sub-frames.  usage: r02_wavefront.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
echo "[1] tests"; timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "path_trace or config1 or config3 or variants_agree or tile_partition_matches or degenerate or textures" > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log; tail -3 $OUT/tests.log
grep -q "rc=0" $OUT/tests.log || exit 1
for scene in teapots cornell; do for m in wavefront wavefront_sort wavefront2 wavefront_sort2 persistent; do
  timeout -k 10 300 python3 bench.py --scene $scene --mode $m --steps 10 --warmup 3 --no-cpu-baseline --no-traversal-only --no-pipelined > $OUT/bench_${scene}_$m.json 2> $OUT/bench_${scene}_$m.err
  python3 -c "
import json; j=json.load(open('$OUT/bench_${scene}_$m.json')); print('$scene $m: ms/frame', j['ms_per_step'], 'Mrays/s', j['value'], 'kernel', j['roofline']['kernel'], j['roofline']['avg_launch_ms'])"
done; done
echo "[3] rocprof teapots wavefront_sort"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_wf -- python3 bench.py --scene teapots --mode wavefront_sort --steps 10 --warmup 3 --no-cpu-baseline --no-traversal-only --no-pipelined > $OUT/trace_wf.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace_wf/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'].split('(')[0][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
