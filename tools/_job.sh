# scratch job for one gpurun call (GPU box); the last content: the round's final validation + records
cd $GRAFT_REPO_ROOT
bash tools/validate_all.sh
bash tools/prof_r04.sh > gpurun_out/prof_r04.log 2>&1
tail -n 42 gpurun_out/prof_r04.log | cut -c1-250 | head -10
