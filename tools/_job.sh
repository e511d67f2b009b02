cd $GRAFT_REPO_ROOT
( time bash tools/prof_r05.sh ) > gpurun_out/prof_r05.log 2>&1
tail -70 gpurun_out/prof_r05.log | cut -c1-700
