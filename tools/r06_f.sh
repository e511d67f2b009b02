cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "quiet or sigma_zero" 2>&1 | tail -6
python3 -m pytest tests/test_gpu_sweep.py -m gpu -q -x 2>&1 | tail -4
for w in d_s50_delay4 d_s50_rn0 d_s24_rdist; do
  for d in "" "--disable NO_QUIET_SF"; do
    python3 bench.py --workload $w $d --no-cpu-baseline --no-pmc --no-single-step --no-collective --no-workloads --detail-out /tmp/x.json 2>/dev/null | tail -n 1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$w', '$d' or 'default', r['kernel'], 'launch_us', r['launch_us'], 'frac', r['frac'])"
  done
done
