#!/bin/bash
# scratch runner (round 3): two-rank rehearsal of both bench workloads on ONE GPU (gloo instead of RCCL, both ranks on device 0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zq; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
export RADISH_FORCE_DEVICE=0 RADISH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for wl in pathtrace restir; do
  say "[$wl] 2 ranks"
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --workload $wl --no-cpu-baseline > $OUT/bench_${wl}_2rank.json 2> $OUT/bench_${wl}_2rank.err; say "   rc=$?"
  tail -1 $OUT/bench_${wl}_2rank.json | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('   ', d['n_gpus'], d['ms_per_step'], d['value'], d.get('parity_check'))" | tee -a $OUT/progress.log
done
say done
