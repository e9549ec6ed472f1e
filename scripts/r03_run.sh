#!/bin/bash
# scratch runner (round 3): k_walk_packet workgroup size (RADISH_PACKET_WG)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zv; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
for g in 256 64 128 512 1024; do
  RADISH_PACKET_WG=$g timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$g -- python3 scripts/pmc_frames.py teapots_lights restir 1920 1080 8 > $OUT/trace_$g.log 2>&1
  python3 - $g <<'P' | tee -a $OUT/progress.log
import csv,glob,sys
g=sys.argv[1]
f=glob.glob(f'/root/repo/gpurun_out/r03zv/trace_{g}/runc/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']; n=n[:n.find('(')][-40:]
    if 'walk_packet' in n: print(f"   wg={g}: {n:40s} avg {float(r['AverageNs'])/1e3:7.1f} min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
P
done
say done
