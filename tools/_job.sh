bash tools/prof_r04.sh > gpurun_out/prof_r04.log 2>&1
tail -n 60 gpurun_out/prof_r04.log
