#!/bin/bash
# round-4 working session D: the fourfold (b) schedule at 2^20 .. 2^22, default bench line
out=gpurun_out/r04d
mkdir -p $out
for lg in 20 21 22; do python3 tools/perf_hostapi_schedule.py $lg BLS12-381 quick > $out/hostapi_schedule_$lg.txt 2>&1; cat $out/hostapi_schedule_$lg.txt; done
python3 bench.py --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print("bench default: ms_per_step", d["ms_per_step"], d["roofline"]["phase_ms"], "pcie", d["extra"]["pcie_inclusive"])
PY
echo all-done
