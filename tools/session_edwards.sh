#!/bin/bash
# twisted Edwards bucket sums for subgroup-trusted BLS12-377 G1 plans: the GPU suite, then the same-box A/B.
set -e
OUT=${1:-gpurun_out/ed}
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "edwards" > $OUT/gpu_tests_ed.txt 2>&1
echo ed-tests-done
timeout -k 10 400 python3 tools/perf_edwards.py 22 > $OUT/perf22.txt 2>&1
timeout -k 10 400 python3 tools/perf_edwards.py 20 > $OUT/perf20.txt 2>&1
timeout -k 10 400 python3 tools/perf_edwards.py 19 > $OUT/perf19.txt 2>&1
echo perf-done
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/gpu_tests.txt 2>&1
echo all-done
