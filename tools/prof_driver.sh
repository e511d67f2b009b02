# usage: bash tools/prof_driver.sh    (GPU box)
# rocprofv3 --kernel-trace --stats of EXACTLY the command the driver runs at round end
# (python3 bench.py --gpus 1 --steps 20 --warmup 5), so that the average duration of the dominant
# kernel can be compared with roofline.launch_us of the line that same run printed.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof_driver
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_driver -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof_driver.log 2>&1
find gpurun_out/prof_driver -name "*kernel_stats.csv" -exec cp {} gpurun_out/stats_driver.csv \;
find gpurun_out/prof_driver -name "*_kernel_trace.csv" -delete
find gpurun_out/prof_driver -name "*.db" -delete
grep '^{"metric"' gpurun_out/prof_driver.log > gpurun_out/bench_under_rocprof_driver.json
cut -c1-400 gpurun_out/bench_under_rocprof_driver.json
head -8 gpurun_out/stats_driver.csv | cut -c1-220
