set -e
mkdir -p gpurun_out/r03o
for r in 1 2; do
  python3 bench.py --config 5 --steps 8 --warmup 2 --kernels-only > gpurun_out/r03o/c5_ahead_prio_$r.json 2> gpurun_out/r03o/c5.err
  MLHIP_SORT_AHEAD_PRIO=0 python3 bench.py --config 5 --steps 8 --warmup 2 --kernels-only > gpurun_out/r03o/c5_ahead_noprio_$r.json 2> gpurun_out/r03o/c5.err
  MLHIP_SORT_AHEAD=0 python3 bench.py --config 5 --steps 8 --warmup 2 --kernels-only > gpurun_out/r03o/c5_inline_$r.json 2> gpurun_out/r03o/c5.err
done
python3 bench.py --config 4 --steps 3 --warmup 1 --kernels-only > gpurun_out/r03o/c4_ahead_prio.json 2> gpurun_out/r03o/c4.err
MLHIP_SORT_AHEAD_PRIO=0 python3 bench.py --config 4 --steps 3 --warmup 1 --kernels-only > gpurun_out/r03o/c4_ahead_noprio.json 2> gpurun_out/r03o/c4.err
MLHIP_SORT_AHEAD=0 python3 bench.py --config 4 --steps 3 --warmup 1 --kernels-only > gpurun_out/r03o/c4_inline.json 2> gpurun_out/r03o/c4.err
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r03o/c*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["ms_per_step"], 3), {k: round(v, 2) for k, v in d["roofline"]["phase_ms"].items() if "tail" not in k})
PY
