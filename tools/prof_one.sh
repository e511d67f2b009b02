# usage: bash tools/prof_one.sh <workload> [extra bench args]   (GPU box; writes gpurun_out/stats_<workload>.csv)
w=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$w -- python3 bench.py --workload $w --no-cpu-baseline --no-single-step "$@" > gpurun_out/prof_$w.log 2>&1
find gpurun_out/prof_$w -name "*kernel_stats.csv" -exec cp {} gpurun_out/stats_$w.csv \;
find gpurun_out/prof_$w -name "*_kernel_trace.csv" -delete
find gpurun_out/prof_$w -name "*.db" -delete
tail -1 gpurun_out/prof_$w.log | cut -c1-150
head -6 gpurun_out/stats_$w.csv | cut -c1-200
