#!/bin/bash
# BN254 pairings on the carry-free lane-pair kernels (default) against the saturated ones (MLHIP_PAIRING_SAT=1): the
# pairing / Gt parity tests both ways, then the same-box A/B of tools/perf_pairing.py.  Usage: bash tools/session_lp28_377.sh [out]
set -e
OUT=${1:-gpurun_out/lp377}
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "pairing or gt_ or Gt or fexp or miller" > $OUT/tests_default.txt 2>&1
echo tests-default-done
MLHIP_PAIRING_SAT=1 timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "pairing or gt_ or Gt or fexp or miller" > $OUT/tests_sat.txt 2>&1
echo tests-sat-done
for i in 1 2; do
  MLHIP_PAIRING_SAT=1 timeout -k 10 300 python3 tools/perf_pairing.py BN254 > $OUT/perf_sat_$i.txt 2>&1
  timeout -k 10 300 python3 tools/perf_pairing.py BN254 > $OUT/perf_lp28_$i.txt 2>&1
done
MLHIP_PERF_CURVE=BN254 timeout -k 10 300 python3 tools/perf_gt_exp.py > $OUT/gtexp_lp28.txt 2>&1 || true
MLHIP_PERF_CURVE=BN254 MLHIP_PAIRING_SAT=1 timeout -k 10 300 python3 tools/perf_gt_exp.py > $OUT/gtexp_sat.txt 2>&1 || true
echo all-done
