#!/bin/bash
# One-call measurement set for the round: bench lines, rocprofv3 kernel stats, PMC passes.  usage: round_measure.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
echo "[1] default bench"; timeout -k 10 400 python3 bench.py 2>&1 | tail -1 > $OUT/bench_default.json; cut -c1-200 $OUT/bench_default.json
for m in mega wavefront wavefront_sort; do echo "[2] cornell $m"; timeout -k 10 300 python3 bench.py --mode $m --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_cornell_$m.json; done
echo "[2b] F=2"; timeout -k 10 300 python3 bench.py --frames-in-flight 2 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_cornell_persistent_f2.json
for m in persistent wavefront_sort mega; do echo "[3] teapots $m"; timeout -k 10 300 python3 bench.py --scene teapots --mode $m --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_teapots_$m.json; done
echo "[3b] config 5 stand-in (1.0 M tris, 3840x2160), one GPU"; timeout -k 10 400 python3 bench.py --scene teasets_1m --width 3840 --height 2160 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_teasets1m_4k_persistent.json
echo "[4] restir"; timeout -k 10 300 python3 scripts/bench_restir.py 2>&1 | tail -2 > $OUT/restir.json
echo "[5] rocprof kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_persistent -- python3 bench.py --no-cpu-baseline > $OUT/trace_persistent.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_restir -- python3 scripts/bench_restir.py > $OUT/trace_restir.log 2>&1
echo "[6] pmc"; for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -d' ' -f1); echo "   pass $n"
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc/$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$n.log 2>&1 || echo "pass $n failed"
done
python3 scripts/pmc_summarize.py $OUT/pmc > $OUT/pmc_summary.txt
echo done
