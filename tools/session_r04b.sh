#!/bin/bash
# round-4 working session B: the -m gpu suite on the product library and on the test build, the kernel timeline of a config-2
# step, A/B of the unrolled chunk kernel, the host-protocol schedules, the BLS12-377 fixed-base Mul on Edwards additions
out=gpurun_out/r04b
mkdir -p $out
export TMPDIR=/tmp
python3 tools/check_codeobj.py mathlib_amd/libmlhip.so mathlib_amd/libmlhip_alt.so > $out/codeobj.txt 2>&1; tail -2 $out/codeobj.txt
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests_product.txt 2>&1; echo "rc $?" >> $out/gpu_tests_product.txt; tail -3 $out/gpu_tests_product.txt
MLHIP_LIB=$PWD/mathlib_amd/libmlhip_alt.so timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests_alt.txt 2>&1; echo "rc $?" >> $out/gpu_tests_alt.txt; tail -3 $out/gpu_tests_alt.txt
rocprofv3 --kernel-trace -d $out/trace_c2 -o t --output-format csv -- python3 bench.py --config 2 --kernels-only --steps 5 --warmup 2 > $out/bench_c2_under_trace.json 2> $out/trace_c2.err
python3 tools/trace_gaps.py $(find $out/trace_c2 -name "*kernel_trace.csv" | head -1) > $out/step_timeline.txt 2>&1
cat $out/step_timeline.txt
find $out -name "*kernel_trace.csv" -size +5M -delete
for rep in 1 2 3; do
  for two in 0 1; do
    MLHIP_CHUNKS_TWO=$two python3 bench.py --config 2 --kernels-only --steps 20 --warmup 5 > $out/chunks_two${two}_$rep.json 2>/dev/null
    python3 - <<PY
import json
d=json.loads(open("$out/chunks_two${two}_$rep.json").read().strip().splitlines()[-1])
p=d["roofline"]["phase_ms"]
print("MLHIP_CHUNKS_TWO=$two rep $rep: ms_per_step %.3f  reduce %.3f  accumulate %.3f  sort %.3f" % (d["ms_per_step"], p["g1_reduce"], p["g1_accumulate"], p["g1_sort"]))
PY
  done
done | tee $out/chunks_two_ab.txt
python3 tools/perf_hostapi_schedule.py 20 > $out/hostapi_schedule_20.txt 2>&1; cat $out/hostapi_schedule_20.txt
python3 tools/perf_hostapi_schedule.py 22 BLS12-381 quick > $out/hostapi_schedule_22.txt 2>&1; cat $out/hostapi_schedule_22.txt
python3 tools/perf_fixed_base_ed.py 20 2>&1 | tee $out/fixed_base_edwards_ab.txt
python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("bench default: ms_per_step", d["ms_per_step"], "pcie", d["extra"]["pcie_inclusive"])
PY
echo all-done
