cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/rows; mkdir -p $o
{
for i in 1 2 3 4 5 6; do
timeout 600 python3 tools/ablate.py run mdpp_image.hip cfg4 numpy shipped
timeout 600 python3 tools/ablate.py run mdpp_image.hip cfg4 numpy shipped disable=NO_IMG_NEARTAB
done
} > $o/ablate_i3.txt 2>&1
cut -c1-200 $o/ablate_i3.txt | grep -v "^$" | awk '{print (NR%2==1?"table   ":"no table") " " $3 " " $4}' | tail -12
