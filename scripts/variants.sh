#!/bin/bash
# usage: scripts_variants.sh <mode> <variant>...   — bench each tuning build of libradish_hip (RADISH_HIP_LIB)
MODE=$1; shift
for v in "$@"; do
  if [ "$v" = "base" ]; then LIB=""; else LIB=$GRAFT_REPO_ROOT/radish_pt_amd/csrc/variants/libradish_hip_$v.so; fi
  R=$(RADISH_HIP_LIB=$LIB timeout -k 10 300 python bench.py --mode $MODE --steps ${STEPS:-10} --warmup ${WARM:-2} --no-cpu-baseline 2>&1 | tail -1)
  echo "$v: $(echo $R | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "Mrays/s  roofline", d["roofline"]["frac"])' 2>/dev/null || echo FAILED $R | cut -c1-300)"
done
