#!/bin/bash
# round 4, shifted-base tables: full GPU suite on the product library, same-box A/Bs (perf_fold.py), a bench line, a short soak
out=gpurun_out/r04g
mkdir -p $out
set -o pipefail
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $out/gpu_tests.txt || exit 1
export MLHIP_PERF_PLAIN_C=0
for a in "BLS12-381 20 20" "BLS12-377 20 20" "BN254 20 20" "BLS12-381 17 20" "BLS12-381 18 20" "BLS12-381 19 20" "BLS12-381 21 20" "BLS12-381 22 20" "BLS12-377 19 20"; do
  timeout -k 10 300 python3 tools/perf_fold.py $a 2>&1 | grep -v amdgpu.ids | tee -a $out/perf_fold.txt || exit 1
done
MLHIP_PERF_GROUP=2 timeout -k 10 300 python3 tools/perf_fold.py BLS12-381 20 20 2>&1 | grep -v amdgpu.ids | tee -a $out/perf_fold.txt || exit 1
unset MLHIP_PERF_PLAIN_C
timeout -k 10 400 python3 bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > $out/bench_config2.json || exit 1
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04g/bench_config2.json").read())
print({k:d[k] for k in ("value","ms_per_step")}); print(json.dumps(d["extra"].get("resident_bases_library_geometry"))[:1500]); print(d["extra"].get("pcie_inclusive"))
PY
timeout -k 10 330 python3 tools/soak.py 300 57 2>&1 | tail -3 | tee $out/soak_seed57.txt
echo all-done
