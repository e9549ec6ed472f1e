#!/bin/bash
# PMC passes over k_walk_persistent.  usage: r02_pmc_walker.sh <tag> <scene>
TAG=$1; SCENE=${2:-cornell}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
i=0
for pass in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
            "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" \
            "GRBM_GUI_ACTIVE TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc$i -- python3 scripts/walker_only.py $SCENE > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('$OUT/pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_walk_persistent' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
launches = 3
for k in sorted(tot):
    print(f"{k:45s} {tot[k]/launches:18.0f} per launch")
PY
