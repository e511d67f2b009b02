# scratch job for one gpurun call (GPU box); the last content: the round's validation
cd $GRAFT_REPO_ROOT
bash tools/validate_all.sh
