mkdir -p gpurun_out/r03q; O=gpurun_out/r03q
V=$GRAFT_REPO_ROOT/radish_pt_amd/csrc/variants
b() { # label lib mode scene
  R1=$(RADISH_HIP_LIB=$2 timeout -k 10 120 python3 bench.py --mode $3 --scene $4 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  echo "$(date +%T) $1 $4 $3: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"])' 2>/dev/null || echo FAILED)" | tee -a $O/progress.log
}
for v in base leaf13 leaf12 leaf18 refill8 refill32 fin8 fin32; do
  if [ $v = base ]; then L=""; else L=$V/libradish_hip_$v.so; fi
  b $v "$L" wavefront_sort2 teapots; b $v "$L" wavefront_sort teapots; b $v "$L" wavefront_sort2 cornell
done
