mkdir -p gpurun_out/r03n; O=gpurun_out/r03n
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "trace_ or restir" > $O/test.log 2>&1; tail -2 $O/test.log
b() { # label env mode scene W H
  R1=$(RADISH_PAIRS=$2 timeout -k 10 120 python3 bench.py --mode $3 --scene $4 --width $5 --height $6 --steps 8 --warmup 3 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  echo "$(date +%T) $1 $4 $5x$6 $3: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"])' 2>/dev/null || echo FAILED)" | tee -a $O/progress.log
}
for M in wavefront_sort2 wavefront_sort persistent; do b threaded 0 $M cornell 1920 1080; b pairs 1 $M cornell 1920 1080; done
for P in 0 1; do
  R1=$(RADISH_PAIRS=$P timeout -k 10 200 python3 bench.py --workload restir --steps 8 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$(date +%T) restir config 4 pairs=$P: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", {k: d["config"].get(k) for k in ("ms_gbuffer_kernels","ms_restir_kernels")}, d.get("roofline",{}).get("frac"))' 2>/dev/null || echo FAILED $R1 | cut -c1-300)" | tee -a $O/progress.log
done
