#!/bin/bash
# One-call measurement set for the round: bench lines, rocprofv3 kernel stats, PMC passes.  usage: round_measure.sh <tag>
# Every step runs under its own timeout and appends a progress line to $OUT/progress.log (gpurun kills silent runs).
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[1] default bench"; timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-260 $OUT/bench_default.json
for m in mega wavefront wavefront_sort wavefront2 wavefront_sort2; do say "[2] cornell $m"; timeout -k 10 300 python3 bench.py --mode $m --steps 10 --warmup 3 --no-cpu-baseline --no-traversal-only --no-pipelined > $OUT/bench_cornell_$m.json 2>/dev/null; done
for m in persistent wavefront wavefront_sort wavefront2 wavefront_sort2 mega; do say "[3] teapots $m"; timeout -k 10 300 python3 bench.py --scene teapots --mode $m --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined > $OUT/bench_teapots_$m.json 2>/dev/null; done
say "[3b] config 5 stand-in (1.0 M tris, 3840x2160), one GPU"; timeout -k 10 400 python3 bench.py --scene teasets_1m --width 3840 --height 2160 --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined > $OUT/bench_teasets1m_4k_persistent.json 2>/dev/null
say "[4] restir split / fused"; timeout -k 10 300 python3 scripts/bench_restir.py 2>/dev/null | tail -2 > $OUT/restir_split.json; RADISH_RESTIR_FUSED=1 timeout -k 10 300 python3 scripts/bench_restir.py 2>/dev/null | tail -2 > $OUT/restir_fused.json
say "[4b] post kernels"; timeout -k 10 300 python3 scripts/bench_post.py > $OUT/post_kernels.json 2>/dev/null
say "[5] rocprof kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_persistent -- python3 bench.py --no-cpu-baseline --no-pipelined > $OUT/trace_persistent.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_restir -- python3 scripts/bench_restir.py > $OUT/trace_restir.log 2>&1
say "[6] pmc"; for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1); say "   pass $n"
  timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc/$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traversal-only --no-pipelined > $OUT/pmc_$n.log 2>&1 || say "pass $n failed"
done
python3 scripts/pmc_summarize.py $OUT/pmc > $OUT/pmc_summary.txt
say done
