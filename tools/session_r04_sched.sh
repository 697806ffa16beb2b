mkdir -p gpurun_out/r04i
export MLHIP_PERF_PLAIN_C=16
for sch in default "3,13" "3,5,8" "2,4,10" "3,4,9" "2,3,4,7" "1,2,4,9" "3,6,7"; do
  if [ "$sch" = default ]; then unset MLHIP_STREAM_SCHEDULE; else export MLHIP_STREAM_SCHEDULE=$sch; fi
  echo "== schedule $sch" | tee -a gpurun_out/r04i/sched.txt
  timeout -k 10 200 python3 tools/perf_fold.py BLS12-381 20 20 2>&1 | grep -v amdgpu | sed "s/create.*| host scalars/| host scalars/" | cut -c1-170 | tee -a gpurun_out/r04i/sched.txt
done
unset MLHIP_STREAM_SCHEDULE
echo "== sort-ahead off" | tee -a gpurun_out/r04i/sched.txt
MLHIP_SORT_AHEAD=0 timeout -k 10 200 python3 tools/perf_fold.py BLS12-381 20 20 2>&1 | grep -v amdgpu | sed "s/create.*| host scalars/| host scalars/" | cut -c1-170 | tee -a gpurun_out/r04i/sched.txt
