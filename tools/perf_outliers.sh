#!/bin/bash
# step-time outliers of the default bench under a knob: N runs each with and without it
# Usage: bash tools/perf_outliers.sh OUTDIR "ENV=val" [runs]
out=$1; knob=$2; runs=${3:-4}
mkdir -p "$out"
for i in $(seq 1 $runs); do
  python3 bench.py --steps 40 --warmup 5 --kernels-only > "$out/a_$i.json" 2> "$out/a_$i.err" || exit 1
  env $knob python3 bench.py --steps 40 --warmup 5 --kernels-only > "$out/b_$i.json" 2> "$out/b_$i.err" || exit 1
done
python3 - "$out" "$knob" <<'PY'
import glob, json, sys
out, knob = sys.argv[1], sys.argv[2]
for tag, name in (("a", "default"), ("b", knob)):
    for f in sorted(glob.glob("%s/%s_*.json" % (out, tag))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        e = d["extra"]
        print("%-28s ms/step %.3f  median %.3f  min %.3f  max %.3f  tail %.3f" % (name, d["ms_per_step"], e["ms_per_step_median"], e["ms_per_step_min"], e["ms_per_step_max"], d["roofline"]["phase_ms"]["g1_host_tail"]))
PY
