#!/bin/bash
# Issue-side counters of the two dominant kernels side by side: k_accumulate28 (bench.py --kernels-only) and
# k_pairing_lp28 (tools/perf_pairing.py).  Usage: bash tools/pmc_issue.sh <outdir>
set -e
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_issue}
mkdir -p $out
C="SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES"
rocprofv3 --kernel-trace --pmc $C -d $out/pairing -o p --output-format csv -- python3 tools/perf_pairing.py BLS12-381 > $out/pairing.log 2>&1 || tail -5 $out/pairing.log
rocprofv3 --kernel-trace --pmc $C -d $out/msm -o p --output-format csv -- python3 bench.py --kernels-only --steps 3 --warmup 1 > $out/msm.log 2>&1 || tail -5 $out/msm.log
python3 - $out <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for d in ("pairing", "msm"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mlhip::", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "pairing_lp28<Bls381, 2" in k or "accumulate28" in k or "chunks_q" in k or "masked" in k:
            print(k, {c: "%.4e" % (sum(x) / len(x)) for c, x in sorted(v.items())})
PY
