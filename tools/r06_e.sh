cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 tools/repro_cfg5.py 2>&1 | grep "^4096 0 0\|^65536 0 0" | cut -c1-100
python3 -m pytest tests/test_gpu_sweep.py -m gpu -q -x -k "sigma_zero" 2>&1 | tail -4
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sigma_zero or quiet" 2>&1 | tail -4
for w in d_s50_rn0 d_s24_rdist c_d2_n0 d_s8_rn0; do
  for d in "" "--disable NO_TRIO"; do
    python3 bench.py --workload $w $d --no-cpu-baseline --no-pmc --no-single-step --no-collective --no-workloads --detail-out /tmp/x.json 2>/dev/null | tail -n 1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$w', '$d' or 'default', r['kernel'], 'launch_us', r['launch_us'], 'frac', r['frac'])"
  done
done
