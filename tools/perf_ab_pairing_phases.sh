for r in 1 2; do
  echo "== in-tree, pairs"; MLHIP_PAIRING_QUAD=0 python3 tools/perf_pairing.py BLS12-381 2>/dev/null | grep batch
  echo "== prev, pairs"; MLHIP_PAIRING_QUAD=0 MLHIP_LIB=$PWD/mathlib_amd/libmlhip_prev.so python3 tools/perf_pairing.py BLS12-381 2>/dev/null | grep batch
done
echo "== in-tree, default (Miller alone on quads)"; python3 tools/perf_pairing.py BLS12-381 2>/dev/null | grep batch
