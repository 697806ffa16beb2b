#!/bin/bash
# same-box A/B of builds of the library on the fused pairing kernel (lane pairs, batch 65 536): the in-tree build
# against MLHIP_LIB=<other .so> ..., alternating, four rounds.  Usage: bash tools/perf_ab_pairing.sh a.so [b.so ...]
for r in 1 2 3 4; do
  echo "== in-tree build"; python3 tools/perf_pairing.py BLS12-381 2>/dev/null | grep "pairing batch"
  for other in "$@"; do
    echo "== $other"; MLHIP_LIB=$PWD/$other python3 tools/perf_pairing.py BLS12-381 2>/dev/null | grep "pairing batch"
  done
done
