#!/bin/bash
# latency form of k_pt_persistent: parity tests that touch it, then per-rank partition times with LOOK = 2 / 4 / 8 variants
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[1] parity"; timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py::test_path_trace_bit_exact tests/test_gpu_configs.py::test_config3_teapots_1080p_wavefront_sort tests/test_gpu_configs.py -x -q -m gpu -k "path_trace_bit_exact or config3 or config1" > $OUT/parity.log 2>&1; say "   rc=$? $(tail -1 $OUT/parity.log)"
for v in default look2 look8; do
  say "[2] partition times, variant $v"
  if [ $v = default ]; then unset RADISH_HIP_LIB; else export RADISH_HIP_LIB=$R/radish_pt_amd/csrc/variants/libradish_hip_$v.so; fi
  timeout -k 10 300 python3 scripts/partition_times.py teapots 1920 1080 persistent_bulk,persistent_latency > $OUT/partition_$v.txt 2>&1; grep '^{"mode' $OUT/partition_$v.txt
done
say done
