#!/bin/bash
# PMC passes over the walk-only kernel, threaded (RADISH_PAIRS=0: k_walk_persistent) and sibling pairs (k_walk_pair), same 8 M incoherent rays.
# Every pass asks for at most three TCP counters (more than the hardware holds makes rocprofv3 abort: round 2's pmc3).  usage: r03_pmc_walker.sh <tag> [scene]
TAG=$1; SCENE=${2:-teapots}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
for P in 0 1; do
  export RADISH_PAIRS=$P
  i=0
  for pass in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
              "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
              "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
              "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
              "FETCH_SIZE" \
              "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/p${P}_pmc$i -- python3 scripts/walker_only.py $SCENE > $OUT/p${P}_pmc$i.log 2>&1 || echo "pairs=$P pass $i failed" | tee -a $OUT/summary.txt
    echo "$(date +%T) pairs=$P pass $i done" >> $OUT/progress.log
  done
  python3 - <<PY | tee -a $OUT/summary.txt
import csv, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob('$OUT/p${P}_pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_walk_' in r['Kernel_Name']:
            tot[(r['Kernel_Name'].split('<')[0].replace('void ', ''), r['Counter_Name'])] += float(r['Counter_Value'])
print("== RADISH_PAIRS=$P, $SCENE, 8 M incoherent class-0 rays, per launch (3 launches)")
for k in sorted(tot):
    print(f"{k[0]:26s} {k[1]:36s} {tot[k]/3:18.0f}")
PY
done
