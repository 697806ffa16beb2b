set -e
export TMPDIR=/tmp
out=gpurun_out/r2g_icache
mkdir -p $out
rocprofv3 -L > $out/counters.txt 2>&1 || true
grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_INSTS_SMEM" $out/counters.txt | sort -u | head -60
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d $out/ic -o p --output-format csv -- python3 tools/perf_pairing.py BLS12-381 > $out/ic.log 2>&1 || tail -5 $out/ic.log
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d $out/ic2 -o p --output-format csv -- python3 bench.py --kernels-only --steps 3 --warmup 1 > $out/ic2.log 2>&1 || tail -5 $out/ic2.log
python3 - <<'PY'
import csv, glob, collections
for d in ("ic","ic2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob("gpurun_out/r2g_icache/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mlhip::", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "pairing" in k or "accumulate28" in k or "chunks" in k:
            print(k, {c: "%.4e" % (sum(x)/len(x)) for c, x in v.items()})
PY
