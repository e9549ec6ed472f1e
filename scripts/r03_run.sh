mkdir -p gpurun_out/r03z7; O=gpurun_out/r03z7
echo skip-tests
b() { # label mode scene
  R1=$(timeout -k 10 120 python3 bench.py --mode $2 --scene $3 --steps 12 --warmup 3 --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only 2>/dev/null | tail -1)
  echo "$(date +%T) $1 $3 $2: $(echo $R1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  frac", d["roofline"]["frac"])' 2>/dev/null || echo FAILED)" | tee -a $O/progress.log
}
for i in 1 2; do b lit0 wavefront_sort2 teapots; b lit0 wavefront2 teapots; b lit0 wavefront_sort teapots; done
b lit0 wavefront_sort2 cornell
