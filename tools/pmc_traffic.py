#!/usr/bin/env python3
"""Turns the rocprofv3 counter CSVs of the HBM-traffic passes into profiles/<name>.json.
Passes (each its own run, as the MI355X guide prescribes -- FETCH_SIZE and WRITE_SIZE do not fit one pass), per config:
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o f --output-format csv -- python3 bench.py --config N --kernels-only --steps 3 --warmup 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir> -o w --output-format csv -- python3 bench.py --config N --kernels-only --steps 3 --warmup 1
Usage: pmc_traffic.py out.json <source_hash> [<commit>] N:<counter_collection.csv> ...   (tools/profile_round.sh makes the CSVs;
<source_hash> = `python -m mathlib_amd.build --source-hash` on the box that took the passes)"""
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
    out_path, source_hash = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    commit = None
    if rest and ":" not in rest[0]:
        commit, rest = rest[0], rest[1:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
    for arg in rest:
        cfg, path = arg.split(":", 1)
        for r in csv.DictReader(open(path)):
            acc[cfg][short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --config N --kernels-only "
        "--steps 3 --warmup 1` (MI355X; tools/profile_round.sh). Values are per launch, in KiB as rocprofv3 reports them, averaged "
        "over the launches of each kernel. FETCH_SIZE is NOT doubled: the x2 gfx950 correction of MI355X_MICROARCH.md applies to "
        "wide coalesced streaming reads, while the accumulation kernels read per-lane 16-byte pieces of randomly gathered rows "
        "(uncalibrated pattern); Infinity-Cache hits are counted by this counter.",
        "source_hash": source_hash,
        "commit": commit,
        "configs": {cfg: {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": max(len(v) for v in d.values())}
                          for k, d in kernels.items() if k.startswith("k_")} for cfg, kernels in sorted(acc.items())},
    }
    json.dump(out, open(out_path, "w"), indent=1)
    for cfg, kernels in out["configs"].items():
        for k, d in kernels.items():
            print(cfg, k, {c: round(v, 1) for c, v in d.items()})


if __name__ == "__main__":
    main()
