# A/B of the sigma-0 draws: default dispatch against MDPP_OPT_NO_SIGMA0 (values formed), same lease
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
o=gpurun_out/r06c; mkdir -p $o
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "beyond_eight" 2>&1 | tail -4
for w in d_s8_rn0 d_s50_rn0 c_d2_n0; do
  for d in "" "--disable NO_SIGMA0"; do
    python3 bench.py --workload $w $d --no-cpu-baseline --no-pmc --no-single-step --no-collective --no-workloads --detail-out $o/x.json 2>/dev/null | tail -n 1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$w', '$d' or 'default', r['kernel'], 'launch_us', r['launch_us'], 'frac', r['frac'])"
  done
done
