#!/usr/bin/env python3
"""Turns the rocprofv3 counter CSVs of the two HBM-traffic passes into profiles/<name>.json.
Passes (each its own run, as the MI355X guide prescribes -- FETCH_SIZE and WRITE_SIZE do not fit one pass):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pairing
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pairing
Usage: pmc_traffic.py <counter_collection.csv> [more CSVs: config-3 passes ...] out.json   (tools/profile_round.sh makes the CSVs)"""
import collections
import csv
import json
import re
import sys


def short(name):
    n = name.split("(")[0]
    n = re.sub(r"^void ", "", n)
    return n.replace("mlhip::", "")


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[1:-1]:
        for r in csv.DictReader(open(path)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --kernels-only --steps 3 "
        "--warmup 1` and the same with `--config 3` (MI355X; tools/profile_round.sh). Values are per launch, in KiB as rocprofv3 reports them, averaged over the "
        "launches of each kernel. FETCH_SIZE is NOT doubled: the x2 gfx950 correction of MI355X_MICROARCH.md applies to wide "
        "coalesced streaming reads, while the accumulation kernels read per-lane 16-byte pieces of randomly gathered rows "
        "(uncalibrated pattern); Infinity-Cache hits are counted by this counter.",
        "kernels": {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items() if k.startswith("k_")},
    }
    json.dump(out, open(sys.argv[-1], "w"), indent=1)
    for k, d in out["kernels"].items():
        print(k, {c: round(v, 1) for c, v in d.items()})


if __name__ == "__main__":
    main()
