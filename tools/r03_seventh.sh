cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r03g; mkdir -p $o
./build/bench_hbm2 2>&1 | grep "bare pattern" > $o/hbm2.txt
python3 tools/ablate.py run mdpp_image.hip cfg4 numpy ahead1 ahead2 ahead2nt ahead1 ahead2 > $o/cfg4.txt 2>&1
cat $o/hbm2.txt $o/cfg4.txt
