# usage: bash tools/prof_r02b.sh   (GPU box) -- records after the lean kernel / nt stores:
#   the driver's bench command (live PMC traffic), rocprofv3 kernel stats of the same command, the offline
#   traffic record, SQ counters of the kernel, one bench line per workload.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r02b; mkdir -p $o
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' > $o/bench_driver_argv.json
rm -rf gpurun_out/prof_cfg2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg2 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $o/bench_under_rocprof.log 2>&1
grep '^{"metric"' $o/bench_under_rocprof.log > $o/cfg2_bench_under_rocprof.json
find gpurun_out/prof_cfg2 -name "*kernel_stats.csv" -exec cp {} $o/cfg2_kernel_stats.csv \;
rm -rf gpurun_out/prof_cfg2
bash tools/pmc_traffic.sh cfg2 512 4 > /dev/null 2>&1; cp gpurun_out/traffic_cfg2.json $o/traffic_cfg2.json
bash tools/pmc_traffic.sh grid 512 4 > /dev/null 2>&1; cp gpurun_out/traffic_grid.json $o/traffic_grid.json
bash tools/pmc_sq.sh cfg2 512 2 > $o/sq_cfg2.txt 2>&1
bash tools/bench_all.sh $o > $o/summary.txt 2>&1
rm -rf gpurun_out/pmc_* gpurun_out/sq_*
cat $o/summary.txt; head -c 1500 $o/bench_driver_argv.json; echo; head -5 $o/cfg2_kernel_stats.csv
