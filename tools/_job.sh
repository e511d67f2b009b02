cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/rows; mkdir -p $o
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "image or cfg4 or polygon" > $o/tests_i.log 2>&1; tail -3 $o/tests_i.log
timeout 600 python3 tools/soak_images.py > $o/soak_images.log 2>&1; tail -2 $o/soak_images.log
{
for i in 1 2 3; do timeout 600 python3 tools/ablate.py run mdpp_image.hip cfg4 numpy base old; done
} > $o/ablate_i2.txt 2>&1
cut -c1-200 $o/ablate_i2.txt | grep -v "^$" | tail -12
