#!/bin/bash
# The round's judged profile artifacts, produced on the GPU box from the repo root:
#   1. python bench.py (default flags the driver uses)                      -> $out/bench_default.json
#   2. rocprofv3 --kernel-trace --stats of `bench.py --kernels-only`        -> $out/stats/  (+ the bench line under rocprof)
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of the same -> $out/pmc_fetch, $out/pmc_write
#   4. the same three for `bench.py --config 3 --kernels-only` (the pairing batch)
# Usage: bash tools/profile_round.sh gpurun_out/r02_profile
set -e
out=${1:-gpurun_out/r02_profile}
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"
rocprofv3 --kernel-trace --stats -d "$out/stats" -o b --output-format csv -- python3 bench.py --kernels-only --steps 10 --warmup 3 > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch" -o f --output-format csv -- python3 bench.py --kernels-only --steps 3 --warmup 1 > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write" -o w --output-format csv -- python3 bench.py --kernels-only --steps 3 --warmup 1 > "$out/pmc_write.json" 2> "$out/pmc_write.err"
rocprofv3 --kernel-trace --stats -d "$out/stats_c3" -o b --output-format csv -- python3 bench.py --config 3 --kernels-only --steps 5 --warmup 2 > "$out/bench_c3_under_rocprof.json" 2> "$out/stats_c3.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch_c3" -o f --output-format csv -- python3 bench.py --config 3 --kernels-only --steps 3 --warmup 1 > "$out/pmc_fetch_c3.json" 2> "$out/pmc_fetch_c3.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write_c3" -o w --output-format csv -- python3 bench.py --config 3 --kernels-only --steps 3 --warmup 1 > "$out/pmc_write_c3.json" 2> "$out/pmc_write_c3.err"
python3 bench.py --config 3 --steps 5 --warmup 2 > "$out/bench_c3.json" 2> "$out/bench_c3.err"
find "$out" -name "*stats*.csv" | head
