mkdir -p gpurun_out/r03t; O=gpurun_out/r03t
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "trace_ or config3 or path_trace" > $O/test.log 2>&1; tail -2 $O/test.log
b() { # label mode scene W H
  R1=$(timeout -k 10 120 python3 bench.py --mode $2 --scene $3 --width $4 --height $5 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  echo "$(date +%T) $1 $3 $4x$5 $2: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"])' 2>/dev/null || echo FAILED)" | tee -a $O/progress.log
}
for M in wavefront_sort2 wavefront2 wavefront_sort persistent; do b masks $M teapots 1920 1080; done
b masks wavefront_sort2 teasets_1m 3840 2160
b masks persistent cornell 1920 1080
