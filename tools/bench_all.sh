# usage: bash tools/bench_all.sh [outdir]   (GPU box): one bench line per workload (numpy and Philox streams), summary on stdout
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/bench_all}; mkdir -p $out
for w in cfg2 cfg2_noise cfg2_irr cfg3 cfg4 cfg5 grid line img_cont; do
  python3 bench.py --workload $w --no-cpu-baseline --no-pmc --no-single-step 2>/dev/null | grep '^{"metric"' > $out/bench_$w.json
done
for w in cfg2 cfg2_noise cfg3 cfg5 grid; do
  python3 bench.py --workload $w --rng philox --no-cpu-baseline --no-pmc --no-single-step 2>/dev/null | grep '^{"metric"' > $out/bench_${w}_philox.json
done
for f in $out/bench_*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], '%.3e' % d['value'], '%.3f' % r['frac'], '%.1f us' % r.get('launch_us', 0), r['kernel'])"; done
