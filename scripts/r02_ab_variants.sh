#!/bin/bash
# A/B of tuning builds (radish_pt_amd/csrc/variants/libradish_hip_<name>.so; "default" = the in-tree library): bench.py's default
# frame (with its parity check) and the teapots frame per variant.  usage: scripts/r02_ab_variants.sh <outdir> <name> [<name> ...]
out=gpurun_out/$1; shift
mkdir -p $out
for v in "$@"; do
  if [ $v = default ]; then unset RADISH_HIP_LIB; else export RADISH_HIP_LIB=radish_pt_amd/csrc/variants/libradish_hip_$v.so; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-traversal-only --no-pipelined > $out/${v}_cornell.json 2> $out/${v}_cornell.err || { echo "$v cornell FAILED" >> $out/summary.txt; tail -3 $out/${v}_cornell.err >> $out/summary.txt; exit 1; }
  timeout -k 10 200 python bench.py --scene teapots --steps 10 --warmup 5 --no-cpu-baseline --no-traversal-only --no-pipelined > $out/${v}_teapots.json 2> $out/${v}_teapots.err || { echo "$v teapots FAILED" >> $out/summary.txt; exit 1; }
  python - >> $out/summary.txt <<PY
import json
a=json.loads(open("$out/${v}_cornell.json").read().strip().splitlines()[-1]); b=json.loads(open("$out/${v}_teapots.json").read().strip().splitlines()[-1])
print("$v", "cornell", a["ms_per_step"], "frac", a["roofline"]["frac"], "parity", a.get("parity_check"), "| teapots", b["ms_per_step"])
PY
done
