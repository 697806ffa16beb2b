#!/bin/bash
# pairing time against batch size on lane pairs: separates the dependent chain (small batches), SIMD sharing (2^15 -> 2^16:
# one -> two waves per SIMD) and memory contention
for lg in 10 13 14 15 16 17; do
  MLHIP_PAIRING_QUAD=0 python3 tools/perf_pairing.py BLS12-381 $((1 << lg)) 2>/dev/null | grep -v amdgpu.ids
done
