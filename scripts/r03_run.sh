mkdir -p gpurun_out/r03z5; O=gpurun_out/r03z5
V=$GRAFT_REPO_ROOT/radish_pt_amd/csrc/variants
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "trace_" 2>&1 | tail -1 | tee -a $O/progress.log
timeout -k 10 300 python scripts/pair_walker_rate.py teapots teasets_1m cornell 2>&1 | grep "pairs\]\|pairs:" | tee -a $O/progress.log
R1=$(timeout -k 10 200 python3 bench.py --workload restir --steps 8 --no-cpu-baseline 2>/dev/null | tail -1); echo "restir config 4: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms")')" | tee -a $O/progress.log
b() { # label lib mode scene
  R1=$(RADISH_HIP_LIB=$2 timeout -k 10 120 python3 bench.py --mode $3 --scene $4 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  echo "$(date +%T) $1 $4 $3: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"])' 2>/dev/null || echo FAILED)" | tee -a $O/progress.log
}
for v in base pw7 base pw7; do
  if [ $v = base ]; then L=""; else L=$V/libradish_hip_$v.so; fi
  b $v "$L" wavefront_sort2 teapots; b $v "$L" wavefront_sort teapots
done
