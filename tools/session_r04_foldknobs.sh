#!/bin/bash
# env-only knobs of the bucket reduction, 2^20 BLS12-381 G1, plain c = 16 plan and shifted-base tables (tools/perf_fold.py): the
# 'reduce' phase is event-timed on the device and does not drift with the box's clocks as the call times do
out=gpurun_out/r04l; mkdir -p $out
run() { echo "== $1" | tee -a $out/knobs2.txt; env $1 MLHIP_PERF_PLAIN_C=16 MLHIP_FOLD_WINDOW=0 timeout -k 10 200 python3 tools/perf_fold.py BLS12-381 20 0 2>&1 | grep "2^20" | sed "s/create.*| resident scalars/| resident scalars/" | cut -c1-200 | tee -a $out/knobs2.txt; }
run "X=0"
run "MLHIP_CHUNK_LOG2=5"
run "MLHIP_CHUNK_LOG2=6"
run "MLHIP_CHUNK_LOG2=5 MLHIP_RED_BLOCK=128"
run "MLHIP_CHUNK_LOG2=4 MLHIP_RED_BLOCK=128"
run "X=1"
