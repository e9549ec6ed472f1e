#!/bin/bash
# scratch runner (round 3): k_pt_persistent<pairs> held to 4 waves per SIMD (128 VGPRs + 156 B of scratch per lane) — a rank's share of the frame
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03zc; mkdir -p $OUT; cd $R
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.log; }
say "[1] default library: persistent, rank shares"
timeout -k 10 300 python3 scripts/partition_times.py teapots 1920 1080 persistent > $OUT/share_default.txt 2>&1; say "   rc=$?"; grep '"mode"' $OUT/share_default.txt | grep -v rows | tee -a $OUT/progress.log
say "[2] 4 waves per SIMD"
RADISH_HIP_LIB=$R/radish_pt_amd/csrc/variants/libradish_hip_pw4.so timeout -k 10 300 python3 scripts/partition_times.py teapots 1920 1080 persistent > $OUT/share_pw4.txt 2>&1; say "   rc=$?"; grep '"mode"' $OUT/share_pw4.txt | grep -v rows | tee -a $OUT/progress.log
say done
