#!/bin/bash
# round-4 working session C: suite on both libraries after the Jacobian host tail / unrolled chunk kernel / (b) schedule fix,
# default bench line, small-MSM table, (b)/(c) at 2^21 and 2^22
out=gpurun_out/r04c
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests_product.txt 2>&1; echo "rc $?" >> $out/gpu_tests_product.txt; tail -3 $out/gpu_tests_product.txt
MLHIP_LIB=$PWD/mathlib_amd/libmlhip_alt.so timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests_alt.txt 2>&1; echo "rc $?" >> $out/gpu_tests_alt.txt; tail -3 $out/gpu_tests_alt.txt
python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("bench default: ms_per_step", d["ms_per_step"], d["roofline"]["phase_ms"], "pcie", d["extra"]["pcie_inclusive"])
PY
python3 tools/perf_small_msm.py > $out/small_msm.txt 2>&1; cat $out/small_msm.txt
python3 tools/perf_hostapi_schedule.py 22 BLS12-381 quick > $out/hostapi_schedule_22.txt 2>&1; cat $out/hostapi_schedule_22.txt
python3 tools/perf_hostapi_schedule.py 21 BLS12-381 quick > $out/hostapi_schedule_21.txt 2>&1; cat $out/hostapi_schedule_21.txt
echo all-done
