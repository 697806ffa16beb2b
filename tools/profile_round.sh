#!/bin/bash
# The round's judged profile artifacts, produced on the GPU box from the repo root:
#   1. python bench.py (default flags the driver uses)                        -> $out/bench_default.json
#   2. per BASELINE config N in 2 3 4 5 (`bench.py --config N --kernels-only`, rocprofv3 directly in front of python3):
#      rocprofv3 --kernel-trace --stats                                       -> $out/stats_cN/  (+ the bench line under rocprof)
#      rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)        -> $out/pmc_fetch_cN, $out/pmc_write_cN
#   3. the full bench lines of configs 3, 4, 5                                -> $out/bench_cN.json
# Usage: bash tools/profile_round.sh gpurun_out/r03_profile ["2 3 4 5"]
# Afterwards, here: python tools/pmc_traffic.py profiles/rNN_pmc_traffic.json $(cat $out/source_hash.txt) $(git rev-parse --short HEAD) \
#                     2:$out/pmc_fetch_c2/..counter_collection.csv 2:$out/pmc_write_c2/.. 3:.. (tools/collect_profiles.py does it)
set -e
out=${1:-gpurun_out/r03_profile}
configs=${2:-2 3 4 5}
mkdir -p "$out"
export TMPDIR=/tmp
python3 -m mathlib_amd.build --source-hash > "$out/source_hash.txt"
python3 bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"
echo "bench default done"
for c in $configs; do
  steps=10; warm=3
  if [ "$c" = 4 ]; then steps=3; warm=1; fi
  rocprofv3 --kernel-trace --stats -d "$out/stats_c$c" -o b --output-format csv -- python3 bench.py --config $c --kernels-only --steps $steps --warmup $warm > "$out/bench_c${c}_under_rocprof.json" 2> "$out/stats_c$c.err"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch_c$c" -o f --output-format csv -- python3 bench.py --config $c --kernels-only --steps 3 --warmup 1 > "$out/pmc_fetch_c$c.json" 2> "$out/pmc_fetch_c$c.err"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write_c$c" -o w --output-format csv -- python3 bench.py --config $c --kernels-only --steps 3 --warmup 1 > "$out/pmc_write_c$c.json" 2> "$out/pmc_write_c$c.err"
  echo "config $c profiled"
  if [ "$c" != 2 ]; then
    python3 bench.py --config $c --steps 5 --warmup 2 > "$out/bench_c$c.json" 2> "$out/bench_c$c.err"
    echo "config $c bench line done"
  fi
done
# BLS12-377 over a fixed SRS (twisted Edwards bucket sums): kernel stats of the A/B tool
rocprofv3 --kernel-trace --stats -d "$out/stats_ed" -o e --output-format csv -- python3 tools/perf_edwards.py 22 > "$out/perf_edwards_under_rocprof.txt" 2> "$out/stats_ed.err" || true
python3 tools/perf_edwards.py 22 > "$out/perf_edwards22.txt" 2>&1 || true
python3 tools/perf_edwards.py 20 > "$out/perf_edwards20.txt" 2>&1 || true
echo "edwards profiled"
# the traces themselves are large: keep the stats and the counter tables only
find "$out" -name "*kernel_trace.csv" -size +20M -delete
find "$out" -name "*stats*.csv"
