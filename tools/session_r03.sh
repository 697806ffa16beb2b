set -e
mkdir -p gpurun_out/r03f
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r03f/gpu_tests.txt 2>&1
echo tests-done
bash tools/profile_round.sh gpurun_out/r03f_profile "2 3 4 5" > gpurun_out/r03f/profile.log 2>&1
echo profile-done
bash tools/pmc_issue.sh gpurun_out/r03f_pmc_issue > gpurun_out/r03f/pmc_issue.txt 2>&1
python3 tools/perf_small_msm.py > gpurun_out/r03f/small_msm.txt 2>&1
python3 tools/perf_min_device_msm.py > gpurun_out/r03f/min_device_msm.txt 2>&1
MLHIP_BENCH_REHEARSAL=1 python3 bench.py --gpus 2 --config 5 --log-n 18 --kernels-only --steps 3 --warmup 1 > gpurun_out/r03f/selflaunch_c5.json 2> gpurun_out/r03f/selflaunch_c5.err
echo all-done
