#!/bin/bash
# A/B of the G2 accumulation kernels on one box: lane pairs split by coordinate (k_accumulate28_kc_seg, MLHIP_G2_KC=1)
# against the component split (k_accumulate28_lp_seg, MLHIP_G2_KC=0).  Usage (GPU box, repo root): bash tools/session_g2_kc.sh [out]
set -e
OUT=${1:-gpurun_out/g2kc}
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1
echo tests-done
for i in 1 2 3; do
  for v in 0 1; do
    MLHIP_G2_KC=$v timeout -k 10 300 python3 bench.py --config 4 --log-n 21 --kernels-only --steps 10 --warmup 3 > $OUT/shard21_kc${v}_$i.json 2> $OUT/shard21_kc${v}_$i.err
  done
done
echo shard-done
for v in 0 1 0 1; do
  MLHIP_G2_KC=$v timeout -k 10 400 python3 bench.py --config 4 --kernels-only --steps 5 --warmup 2 > $OUT/n24_kc${v}_$RANDOM.json 2>> $OUT/n24.err
done
echo all-done
