#!/bin/bash
# Threshold sweep of the walk-only kernels (k_walk_persistent, k_wf_trace, k_gbuffer_persistent) with variant builds:
# traversal-only figure of bench.py's default frame, bench_restir.py, the teapots wavefront frame.  usage: r02_tune_walker.sh <outdir> <variant>...
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  if [ $v = default ]; then unset RADISH_HIP_LIB; else export RADISH_HIP_LIB=radish_pt_amd/csrc/variants/libradish_hip_$v.so; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined > $out/${v}_cornell.json 2>/dev/null || { echo "$v FAILED" >> $out/summary.txt; exit 1; }
  timeout -k 10 200 python bench.py --scene teapots --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined > $out/${v}_teapots.json 2>/dev/null || exit 1
  timeout -k 10 200 python bench.py --scene teapots --mode wavefront2 --steps 10 --warmup 3 --no-cpu-baseline --no-traversal-only --no-pipelined > $out/${v}_wf2.json 2>/dev/null || exit 1
  timeout -k 10 200 python scripts/bench_restir.py 2>/dev/null | head -1 > $out/${v}_restir.json || exit 1
  python - >> $out/summary.txt <<PY
import json
L=lambda f: json.loads(open(f).read().strip().splitlines()[-1])
a=L("$out/${v}_cornell.json"); b=L("$out/${v}_teapots.json"); c=L("$out/${v}_wf2.json"); d=L("$out/${v}_restir.json")
print("%-8s cornell %.3f trav %.4f (%.3f ms) | teapots %.3f trav %.4f | wf2 %.3f | restir %.3f gb %.3f rs %.3f" % ("$v", a["ms_per_step"], a["roofline"]["traversal_only"]["frac"], a["roofline"]["traversal_only"]["ms"], b["ms_per_step"], b["roofline"]["traversal_only"]["frac"], c["ms_per_step"], d["ms_frame_wall"], d["ms_gbuffer_kernel"], d["ms_restir_kernels"]))
PY
done
