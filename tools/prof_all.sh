set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in cfg4 cfg3 cfg5 cfg2_noise; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -- python3 bench.py --workload $w --no-cpu-baseline --no-single-step > gpurun_out/prof_$w.log 2>&1
  find gpurun_out/prof_$w -name "*kernel_stats.csv" -exec cp {} gpurun_out/stats_$w.csv \;
  find gpurun_out/prof_$w -name "*_kernel_trace.csv" -delete
  find gpurun_out/prof_$w -name "*.db" -delete
done
ls -la gpurun_out
