mkdir -p gpurun_out/r03o; O=gpurun_out/r03o
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "gbuffer or restir" > $O/test.log 2>&1; tail -2 $O/test.log
for P in 0 1; do
  echo "== gbuffer_times RADISH_PAIRS=$P" | tee -a $O/progress.log
  RADISH_PAIRS=$P timeout -k 10 120 python3 scripts/gbuffer_times.py 2>&1 | grep "teapots_camera\|cam 0\|cam 3" | tee -a $O/progress.log
  R1=$(RADISH_PAIRS=$P timeout -k 10 200 python3 bench.py --workload restir --steps 8 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$(date +%T) restir config 4 pairs=$P: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms")' 2>/dev/null || echo FAILED $R1 | cut -c1-300)" | tee -a $O/progress.log
done
for P in 1 0; do echo "== partition times RADISH_PAIRS=$P" | tee -a $O/progress.log; RADISH_PAIRS=$P timeout -k 10 300 python scripts/partition_times.py teapots 1920 1080 persistent,wavefront_sort2,wavefront_sort 2>&1 | grep -v "amdgpu.ids\|scene" | tee -a $O/progress.log; done
