mkdir -p gpurun_out/r03z; O=gpurun_out/r03z
V=$GRAFT_REPO_ROOT/radish_pt_amd/csrc/variants
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "workgroup or gbuffer or restir or trace_" > $O/test.log 2>&1; tail -2 $O/test.log | tee -a $O/progress.log
for v in subs1 subs2 base subs8; do
  if [ $v = base ]; then L=""; else L=$V/libradish_hip_$v.so; fi
  echo "== $v" | tee -a $O/progress.log
  RADISH_HIP_LIB=$L timeout -k 10 200 python scripts/gbuffer_4k.py 2>&1 | grep gbuffer | tee -a $O/progress.log
  R1=$(RADISH_HIP_LIB=$L timeout -k 10 200 python3 bench.py --workload restir --steps 8 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "restir config 4: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms")' 2>/dev/null || echo FAILED)" | tee -a $O/progress.log
done
