#!/bin/bash
# one-stream kernel stats of the bf16 (16-bit storage) leg: rocprofv3 --kernel-trace of bench.py --precision bf16, GLF_STREAMS=0
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bf16
rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
GLF_STREAMS=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT -o bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 5 --warmup 2 \
  --no-exact-f32 --no-config3 --no-bf16 --no-cpu-baseline --no-other-mode > $GRAFT_REPO_ROOT/gpurun_out/prof_bf16.log 2>&1
python3 $GRAFT_REPO_ROOT/profiles/kstats_db.py $OUT/bf16_results.db 11 > $GRAFT_REPO_ROOT/gpurun_out/bf16_one_stream_kernel_stats.csv
