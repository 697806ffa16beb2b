#!/bin/bash
# PMC passes over the pairing kernels (tools/perf_pairing.py, batch 65 536): SQ issue counters, scratch / memory
# instruction counts, FETCH_SIZE and WRITE_SIZE -- each in its own rocprofv3 run, kernel trace only (no other domains).
# Usage (on the GPU box, from the repo root): bash tools/pmc_pairing.sh <outdir> [env assignments for the run]
set -e
out=${1:-gpurun_out/pmc_pairing}
mkdir -p "$out"
export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d "$out/$name" -o p --output-format csv -- python3 tools/perf_pairing.py BLS12-381 > "$out/$name.log" 2>&1
}
run sq SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run mem SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$out" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mlhip::", "")
        if "pairing" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as f:
    for k, d in sorted(acc.items()):
        line = k + ": " + ", ".join("%s=%.4e" % (c, sum(v) / len(v)) for c, v in sorted(d.items()))
        print(line)
        f.write(line + "\n")
PY
