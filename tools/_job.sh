cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/rows; mkdir -p $o
SQ_MORE=1 bash tools/pmc_sq.sh cfg4 512 2 > $o/sq_cfg4.txt 2>&1; grep -A40 "obs_fast" $o/sq_cfg4.txt | head -60
