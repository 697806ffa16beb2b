#!/bin/bash
# in-tree build against MLHIP_LIB=$1 on the pairing kernels of the three curves (batch 65 536), two rounds
for r in 1 2; do
  for c in BLS12-381 BN254 BLS12-377; do
    echo "== in-tree $c"; MLHIP_PAIRING_QUAD=0 python3 tools/perf_pairing.py $c 2>/dev/null | grep batch
    echo "== $1 $c"; MLHIP_PAIRING_QUAD=0 MLHIP_LIB=$PWD/$1 python3 tools/perf_pairing.py $c 2>/dev/null | grep batch
  done
done
