#!/bin/bash
# round-4 working session E: one pairing per quad of lanes on BLS12-377 and BN254 -- parity, then quads against lane pairs
out=gpurun_out/r04e
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_driver_gpu.py tests/test_cpp_driver.py -x -q -m gpu -k "pairing or gt_exp or Gt or final_exp or miller or cpp" > $out/pairing_tests.txt 2>&1; echo "rc $?" >> $out/pairing_tests.txt; tail -4 $out/pairing_tests.txt
timeout -k 10 400 python3 tools/perf_pairing_quad.py BN254 2>&1 | tee $out/pairing_quad_bn254.txt
timeout -k 10 300 python3 tools/perf_latency.py 2>&1 | tee $out/latency.txt
echo all-done
