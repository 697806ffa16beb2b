set -e
mkdir -p gpurun_out/r03p
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "msm" > gpurun_out/r03p/tests.txt 2>&1
echo tests-ok
for r in 1 2 3; do
  python3 bench.py --steps 30 --warmup 5 --kernels-only > gpurun_out/r03p/a_$r.json 2>/dev/null
  MLHIP_LIB=$PWD/mathlib_amd/libmlhip_prev.so python3 bench.py --steps 30 --warmup 5 --kernels-only > gpurun_out/r03p/b_$r.json 2>/dev/null
done
python3 bench.py --config 5 --steps 8 --warmup 2 --kernels-only > gpurun_out/r03p/c5_a.json 2>/dev/null
MLHIP_LIB=$PWD/mathlib_amd/libmlhip_prev.so python3 bench.py --config 5 --steps 8 --warmup 2 --kernels-only > gpurun_out/r03p/c5_b.json 2>/dev/null
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r03p/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["roofline"]["phase_ms"].items() if "tail" not in k})
PY
