cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03f; mkdir -p $o
./build/bench_hbm2 > $o/hbm2.txt 2>&1
bash tools/pmc_sq.sh line 512 2 > $o/sq_line.txt 2>&1
bash tools/pmc_sq.sh cfg5 512 2 rng=philox > $o/sq_cfg5_philox.txt 2>&1
bash tools/pmc_sq.sh cfg2_noise 512 2 rng=philox > $o/sq_cfg2_noise_philox.txt 2>&1
bash tools/pmc_sq.sh cfg4 512 2 > $o/sq_cfg4.txt 2>&1
cat $o/hbm2.txt; cat $o/sq_line.txt $o/sq_cfg5_philox.txt $o/sq_cfg2_noise_philox.txt $o/sq_cfg4.txt
