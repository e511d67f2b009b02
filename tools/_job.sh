cd $GRAFT_REPO_ROOT
bash tools/prof_r05.sh > gpurun_out/r05p_summary.txt 2>&1
tail -60 gpurun_out/r05p_summary.txt | cut -c1-1000
