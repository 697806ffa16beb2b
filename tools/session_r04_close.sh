#!/bin/bash
# The closing GPU session of round 4, in parts that fit one gpurun call each (20 minutes at most):
#   A  the whole -m gpu suite on the product library and on the test build (MLHIP_LIB=libmlhip_alt.so)
#   B  configs 2, 3, 5: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes, THEN profiles/r04_pmc_traffic.json is made on the
#      box (so that every bench line printed after it carries `traffic`), THEN the default bench line and the lines of 3 and 5
#   C  config 4: the same order
#   D  kernel stats of the shifted-base-table path (tools/perf_fold.py under rocprofv3)
# Usage (GPU box, repo root): MLHIP_COMMIT=<short hash> bash tools/session_r04_close.sh A|B|C|D
out=gpurun_out/r04z
mkdir -p $out
export TMPDIR=/tmp
commit=${MLHIP_COMMIT:-worktree}
hash=$(python3 -m mathlib_amd.build --source-hash)
echo "$hash" > $out/source_hash.txt
profile_configs() {
  for c in $1; do
    steps=10; warm=3
    if [ "$c" = 4 ]; then steps=3; warm=1; fi
    rocprofv3 --kernel-trace --stats -d "$out/stats_c$c" -o b --output-format csv -- python3 bench.py --config $c --kernels-only --steps $steps --warmup $warm > "$out/bench_c${c}_under_rocprof.json" 2> "$out/stats_c$c.err"
    echo "config $c: kernel stats done"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/pmc_fetch_c$c" -o f --output-format csv -- python3 bench.py --config $c --kernels-only --steps 3 --warmup 1 > "$out/pmc_fetch_c$c.json" 2> "$out/pmc_fetch_c$c.err"
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/pmc_write_c$c" -o w --output-format csv -- python3 bench.py --config $c --kernels-only --steps 3 --warmup 1 > "$out/pmc_write_c$c.json" 2> "$out/pmc_write_c$c.err"
    echo "config $c: counter passes done"
  done
  find "$out" -name "*kernel_trace.csv" -size +20M -delete
}
make_traffic() {  # every counter table taken so far (this part's and, merged back by an earlier call, none: parts are independent boxes)
  args=""
  for c in 2 3 4 5; do
    for k in fetch write; do
      for f in $(find "$out/pmc_${k}_c$c" -name "*counter_collection.csv" 2>/dev/null); do args="$args $c:$f"; done
    done
  done
  python3 tools/pmc_traffic.py profiles/r04_pmc_traffic_$1.json "$hash" "$commit" $args > "$out/pmc_traffic_$1.txt" 2>&1
  cp profiles/r04_pmc_traffic_$1.json "$out/"
  # bench.py reads profiles/r04_pmc_traffic.json: on this box, the part's own file
  cp profiles/r04_pmc_traffic_$1.json profiles/r04_pmc_traffic.json
}
case "$1" in
A)
  python3 tools/check_codeobj.py mathlib_amd/libmlhip.so mathlib_amd/libmlhip_alt.so > $out/codeobj.txt 2>&1; tail -2 $out/codeobj.txt
  timeout -k 10 700 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests_product.txt 2>&1; echo "rc $?" >> $out/gpu_tests_product.txt; tail -3 $out/gpu_tests_product.txt
  MLHIP_LIB=$PWD/mathlib_amd/libmlhip_alt.so timeout -k 10 700 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests_alt.txt 2>&1; echo "rc $?" >> $out/gpu_tests_alt.txt; tail -3 $out/gpu_tests_alt.txt
  ;;
B)
  profile_configs "2 3 5"
  make_traffic B
  python3 bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "default bench line done"
  for c in 3 5; do python3 bench.py --config $c --steps 5 --warmup 2 > "$out/bench_c$c.json" 2> "$out/bench_c$c.err"; echo "config $c bench line done"; done
  ;;
C)
  profile_configs "4"
  make_traffic C
  python3 bench.py --config 4 --steps 5 --warmup 2 > "$out/bench_c4.json" 2> "$out/bench_c4.err"; echo "config 4 bench line done"
  ;;
D)
  # the shifted-base-table path (msm_fold.h): kernel stats of tools/perf_fold.py (plain table and tables, alternating), 2^20 BLS12-381 G1
  rocprofv3 --kernel-trace --stats -d "$out/stats_fold" -o t --output-format csv -- python3 tools/perf_fold.py BLS12-381 20 20 > "$out/perf_fold_under_rocprof.txt" 2> "$out/stats_fold.err"
  echo "fold kernel stats done"
  ;;
esac
echo part-$1-done
