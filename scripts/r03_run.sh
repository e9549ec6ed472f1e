mkdir -p gpurun_out/r03z3; O=gpurun_out/r03z3
V=$GRAFT_REPO_ROOT/radish_pt_amd/csrc/variants
for v in base wg512 wg256; do
  if [ $v = base ]; then L=""; else L=$V/libradish_hip_$v.so; fi
  echo "== $v" | tee -a $O/progress.log
  RADISH_HIP_LIB=$L timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "workgroup or gbuffer_kernels" 2>&1 | tail -1 | tee -a $O/progress.log
  RADISH_HIP_LIB=$L timeout -k 10 200 python scripts/gbuffer_4k.py 2>&1 | grep gbuffer | tee -a $O/progress.log
  R1=$(RADISH_HIP_LIB=$L timeout -k 10 200 python3 bench.py --workload restir --steps 8 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "restir config 4: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms")' 2>/dev/null || echo FAILED)" | tee -a $O/progress.log
done
RADISH_HIP_LIB=$V/libradish_hip_wg256.so timeout -k 10 300 python scripts/partition_times_restir.py teasets_1m 3840 2160 2>&1 | grep '"tile": 128' | grep -v rows | tee -a $O/progress.log
