#!/bin/bash
# rocprofv3 passes of the shifted-base-table path (tools/run_tables_steps.py): kernel trace -> step timeline; FETCH_SIZE and
# WRITE_SIZE in their own runs -> bytes per launch of the accumulation kernel on the 13-row table
out=gpurun_out/r04q; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 tools/run_tables_steps.py 8 > $out/trace.txt 2> $out/trace.err; echo trace done
python3 tools/trace_gaps.py $(find $out/trace -name "*kernel_trace.csv" | head -1) k_coarse_hist 2 > $out/timeline.txt 2>&1; tail -3 $out/timeline.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o f --output-format csv -- python3 tools/run_tables_steps.py 4 > $out/fetch.txt 2> $out/fetch.err; echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o w --output-format csv -- python3 tools/run_tables_steps.py 4 > $out/write.txt 2> $out/write.err; echo write done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04q/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void mlhip::", "").replace("mlhip::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/r04q/pmc_tables.txt", "w") as o:
    o.write("# per launch, KiB as rocprofv3 reports them (FETCH_SIZE not doubled, as in r04_pmc_traffic.json): tools/run_tables_steps.py 4\n")
    for k, d in sorted(acc.items(), key=lambda kv: -sum(sum(v) / len(v) for v in kv[1].values())):
        o.write("%-60s %s\n" % (k[:60], "  ".join("%s %.1f (%d launches)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(d.items()))))
print(open("gpurun_out/r04q/pmc_tables.txt").read()[:1500])
PY
find $out -name "*kernel_trace.csv" -size +5M -delete
