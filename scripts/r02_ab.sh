#!/bin/bash
# A/B of a tuning build against the default library.  usage: r02_ab.sh <tag> <variant-name> [pytest -k expression]
TAG=$1; VAR=$2; KEXPR=${3:-"trace_closest or trace_occluded or trace_empty or restir_sequence"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
V=$R/radish_pt_amd/csrc/variants/libradish_hip_$VAR.so
echo "[1] correctness of the variant"; RADISH_HIP_LIB=$V timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$KEXPR" > $OUT/tests_$VAR.log 2>&1; echo "rc=$?" >> $OUT/tests_$VAR.log; tail -3 $OUT/tests_$VAR.log
grep -q "rc=0" $OUT/tests_$VAR.log || exit 1
for lib in base $VAR; do
  if [ $lib = base ]; then unset RADISH_HIP_LIB; else export RADISH_HIP_LIB=$V; fi
  echo "[2] walker rate ($lib)"; timeout -k 10 300 python3 scripts/walker_rate.py > $OUT/walker_$lib.txt 2>&1; grep "lane refill" $OUT/walker_$lib.txt
  echo "[3] bench ($lib)"; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pipelined > $OUT/bench_$lib.json 2> $OUT/bench_$lib.err; python3 -c "
import json,sys; j=json.load(open('$OUT/bench_$lib.json')); print('  ms/frame', j['ms_per_step'], 'frac', j['roofline']['frac'], 'traversal_only', j['roofline'].get('traversal_only',{}).get('frac'), j['roofline'].get('traversal_only',{}).get('ms'))"
  echo "[4] restir ($lib)"; timeout -k 10 300 python3 scripts/bench_restir.py 2>&1 | tail -2 | head -1 > $OUT/restir_$lib.json; python3 -c "
import json; j=json.load(open('$OUT/restir_$lib.json')); print('  frame', j['ms_frame_wall'], 'gbuffer', j['ms_gbuffer_kernel'], 'restir', j['ms_restir_kernels'])"
done
