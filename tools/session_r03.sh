#!/bin/bash
# the round's closing GPU session: the whole -m gpu suite at the last code commit, the judged profiles, the PMC passes,
# the small-MSM / MinDeviceMSM / power tables.  Usage (GPU box, repo root): bash tools/session_r03.sh
set -e
mkdir -p gpurun_out/r03z
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r03z/gpu_tests.txt 2>&1
echo tests-done
bash tools/profile_round.sh gpurun_out/r03z_profile "2 3 4 5" > gpurun_out/r03z/profile.log 2>&1
echo profile-done
bash tools/pmc_issue.sh gpurun_out/r03z_pmc_issue > gpurun_out/r03z/pmc_issue.txt 2>&1
bash tools/pmc_pairing.sh gpurun_out/r03z_pmc_pairing > gpurun_out/r03z/pmc_pairing.txt 2>&1
python3 tools/perf_small_msm.py > gpurun_out/r03z/small_msm.txt 2>&1
python3 tools/perf_pairing.py BLS12-381 > gpurun_out/r03z/pairing_phases.txt 2>&1
MLHIP_PAIRING_QUAD=0 python3 tools/perf_latency.py > gpurun_out/r03z/latency_pairs.txt 2>&1 || true
python3 tools/perf_latency.py > gpurun_out/r03z/latency.txt 2>&1 || true
echo all-done
