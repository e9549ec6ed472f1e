mkdir -p gpurun_out/r03r; O=gpurun_out/r03r
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "trace_ or config3 or config1 or config5_teasets_4k_path or path_trace" > $O/test.log 2>&1; tail -2 $O/test.log
timeout -k 10 300 python scripts/pair_walker_rate.py teapots teasets_1m cornell > $O/rate.log 2>&1; grep -v amdgpu $O/rate.log
b() { # label env mode scene W H
  R1=$(RADISH_PAIRS=$2 timeout -k 10 120 python3 bench.py --mode $3 --scene $4 --width $5 --height $6 --steps 8 --warmup 3 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  echo "$(date +%T) $1 $4 $5x$6 $3: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"])' 2>/dev/null || echo FAILED)" | tee -a $O/progress.log
}
for M in wavefront_sort2 wavefront2 wavefront_sort persistent; do b pairs 1 $M teapots 1920 1080; done
for M in wavefront_sort2 persistent; do b pairs 1 $M teasets_1m 3840 2160; done
for M in wavefront_sort2 persistent; do b pairs 1 $M cornell 1920 1080; done
