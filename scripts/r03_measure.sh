#!/bin/bash
# Round 3 measurement set: GPU suite, the default bench line (config 3 + sub-records), rocprofv3 kernel traces, PMC passes on the
# teapots frame (wavefront + sort, three sub-frames) and on the 1-M-triangle scene at 4K.  usage: r03_measure.sh <tag> [notests]
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
if [ "$2" != notests ]; then say "[0] gpu tests"; timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; say "   rc=$? $(tail -1 $OUT/gpu_tests.log)"; fi
say "[1] default bench"; timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; say "   rc=$?"; cut -c1-300 $OUT/bench_default.json
say "[2] other structures on config 3"; for m in wavefront2 persistent wavefront_sort; do timeout -k 10 200 python3 bench.py --mode $m --steps 10 --no-cpu-baseline --no-pipelined --no-configs > $OUT/bench_teapots_$m.json 2>/dev/null; done
say "[3] restir workload"; timeout -k 10 300 python3 bench.py --workload restir --steps 8 > $OUT/bench_restir.json 2> $OUT/bench_restir.err; say "   rc=$?"
say "[4] rocprof kernel trace: default command (timed frames + counting pass), and 6 plain frames"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 bench.py --no-cpu-baseline --no-pipelined --no-configs --no-traversal-only > $OUT/trace_default.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_frames -- python3 scripts/pmc_frames.py teapots wavefront_sort2 1920 1080 6 > $OUT/trace_frames.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_restir -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 8 > $OUT/trace_restir.log 2>&1
pmc() { # dir scene mode W H frames
  local D=$1; shift
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    n=$(echo $pass | cut -d' ' -f1); say "   $D pass $n"
    timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$D/$n -- python3 scripts/pmc_frames.py "$@" > $OUT/${D}_$n.log 2>&1 || say "   pass $n failed"
  done
  python3 scripts/pmc_per_frame.py $OUT/$D ${@: -1} > $OUT/${D}_summary.txt
}
say "[5] pmc teapots wavefront_sort2"; pmc pmc_teapots teapots wavefront_sort2 1920 1080 4
say "[6] pmc teasets_1m persistent 4K"; pmc pmc_teasets teasets_1m persistent 3840 2160 2
say "[7] pmc teapots persistent"; pmc pmc_teapots_persistent teapots persistent 1920 1080 4
say done
