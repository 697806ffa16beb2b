#!/bin/bash
# soak on the final code of round 4 (tools/soak.py: random MSMs / pairing batches / fixed-base products against the C oracle),
# then -- LAST step of the call, one wave, once -- the round-3 form of the Jacobian table helper, with its stderr kept
out=gpurun_out/r04s
mkdir -p $out
timeout -k 10 900 python3 tools/soak.py 840 41 > $out/soak_seed41.txt 2>&1; echo "rc $?" >> $out/soak_seed41.txt; tail -4 $out/soak_seed41.txt
(cd $out && timeout -k 10 60 ../../tools/jac_repro 64 0 --run-faulting-form > faulting_form_stdout.txt 2> faulting_form_stderr.txt; echo "exit $?" >> faulting_form_stdout.txt)
cat $out/faulting_form_stdout.txt; tail -12 $out/faulting_form_stderr.txt
echo all-done
