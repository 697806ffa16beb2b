#!/bin/bash
# Where does the fused pairing kernel wait?  Average VMEM latency (SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM), wait and busy
# cycles, beside the G1 accumulation kernel.  Each counter group is its own rocprofv3 run (kernel trace only).
# Usage: bash tools/pmc_waits.sh <outdir>
set -e
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_waits}
mkdir -p $out
rocprofv3 -L > $out/counters_available.txt 2>&1 || true
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $out/p_$name -o p --output-format csv -- python3 tools/perf_pairing.py BLS12-381 > $out/p_$name.log 2>&1 || tail -3 $out/p_$name.log
  rocprofv3 --kernel-trace --pmc "$@" -d $out/m_$name -o p --output-format csv -- python3 bench.py --kernels-only --steps 3 --warmup 1 > $out/m_$name.log 2>&1 || tail -3 $out/m_$name.log
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run b SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
run c SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVES
run d GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
python3 - $out <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for d, want in (("p", ("pairing_lp28<Bls381, 0", "pairing_lp28<Bls381, 1", "pairing_lp28<Bls381, 2")), ("m", ("accumulate28", "chunks_q28"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob("%s/%s_*/**/*counter_collection.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mlhip::", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if any(w in k for w in want):
            print(k, {c: "%.4e" % (sum(x) / len(x)) for c, x in sorted(v.items())})
PY
