#!/bin/bash
# the closing GPU session of round 3 (second session of the round), in two gpurun calls (20 minutes each at most):
#   part A: the whole -m gpu suite at the last code commit, the judged profiles of configs 2, 3, 5, the Edwards kernels
#   part B: config 4's profiles, the issue / pairing counter passes, the small-MSM, pairing-phase and latency tables
# Usage (GPU box, repo root): bash tools/session_r03b.sh A|B
set -e
mkdir -p gpurun_out/r03y
if [ "$1" = A ]; then
  timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > gpurun_out/r03y/gpu_tests.txt 2>&1
  echo tests-done
  bash tools/profile_round.sh gpurun_out/r03y_profile "2 3 5" > gpurun_out/r03y/profile_A.log 2>&1
  echo profile-done
else
  bash tools/profile_round.sh gpurun_out/r03y_profile4 "4" > gpurun_out/r03y/profile_B.log 2>&1
  echo profile4-done
  bash tools/pmc_issue.sh gpurun_out/r03y_pmc_issue > gpurun_out/r03y/pmc_issue.txt 2>&1
  bash tools/pmc_pairing.sh gpurun_out/r03y_pmc_pairing > gpurun_out/r03y/pmc_pairing.txt 2>&1
  python3 tools/perf_small_msm.py > gpurun_out/r03y/small_msm.txt 2>&1
  python3 tools/perf_pairing.py BLS12-381 > gpurun_out/r03y/pairing_phases.txt 2>&1
  python3 tools/perf_latency.py > gpurun_out/r03y/latency.txt 2>&1 || true
  python3 tools/perf_scalar_mul.py > gpurun_out/r03y/scalar_mul.txt 2>&1 || true
fi
echo all-done
