#!/usr/bin/env python3
"""After `bash tools/profile_round.sh gpurun_out/<dir>` has run on the GPU box: copies the judged summaries into profiles/
(tracked) under the round's prefix and builds the PMC traffic file.
Usage: python tools/collect_profiles.py gpurun_out/r03_profile r03"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    cp = lambda a, b: (shutil.copy(a, os.path.join(prof, b)), print("profiles/" + b))  # noqa: E731
    if os.path.exists(os.path.join(src, "bench_default.json")):
        cp(os.path.join(src, "bench_default.json"), "%s_final_bench.json" % prefix)
    pmc_args = []
    for c in "2345":
        for f in glob.glob(os.path.join(src, "stats_c" + c, "**", "*kernel_stats.csv"), recursive=True):
            cp(f, "%s_bench_config%s_kernel_stats.csv" % (prefix, c))
        for name, dst in (("bench_c%s_under_rocprof.json" % c, "%s_bench_config%s_under_rocprof.json" % (prefix, c)),
                          ("bench_c%s.json" % c, "%s_bench_config%s.json" % (prefix, c))):
            if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)):
                cp(os.path.join(src, name), dst)
        for kind in ("fetch", "write"):
            for f in glob.glob(os.path.join(src, "pmc_%s_c%s" % (kind, c), "**", "*counter_collection.csv"), recursive=True):
                pmc_args.append("%s:%s" % (c, f))
    for f in glob.glob(os.path.join(src, "stats_ed", "**", "*kernel_stats.csv"), recursive=True):
        cp(f, "%s_edwards_kernel_stats.csv" % prefix)
    if pmc_args:
        sh = open(os.path.join(src, "source_hash.txt")).read().strip()
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"),
                               os.path.join(prof, "%s_pmc_traffic.json" % prefix), sh, commit] + pmc_args)


if __name__ == "__main__":
    main()
