# usage: bash tools/prof_r02.sh   (GPU box)  -- round-2 records for profiles/:
#   bench lines of every workload (numpy streams) and of the Philox variants, rocprofv3 kernel stats of the
#   Philox runs, PMC traffic of cfg5 / cfg2_noise / cfg2 with Philox streams.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for w in cfg2_noise cfg3 cfg4 cfg5 grid cfg2_irr line img_cont; do
  python3 bench.py --workload $w --no-cpu-baseline --no-pmc 2>/dev/null | grep '^{"metric"' > gpurun_out/r02/bench_$w.json
done
for w in cfg2 cfg2_noise cfg3 cfg5 grid; do
  python3 bench.py --workload $w --rng philox --no-cpu-baseline --no-pmc 2>/dev/null | grep '^{"metric"' > gpurun_out/r02/bench_${w}_philox.json
done
for w in cfg5 cfg2_noise; do
  rm -rf gpurun_out/prof_${w}_philox
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${w}_philox -- python3 bench.py --workload $w --rng philox --no-cpu-baseline --no-single-step > gpurun_out/prof_${w}_philox.log 2>&1
  find gpurun_out/prof_${w}_philox -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02/stats_${w}_philox.csv \;
  rm -rf gpurun_out/prof_${w}_philox
done
for w in cfg5 cfg2_noise cfg2; do
  bash tools/pmc_traffic.sh $w 512 4 rng=philox > /dev/null 2>&1
  cp gpurun_out/traffic_$w.json gpurun_out/r02/traffic_${w}_philox.json
done
bash tools/pmc_traffic.sh cfg5 512 4 > /dev/null 2>&1; cp gpurun_out/traffic_cfg5.json gpurun_out/r02/traffic_cfg5.json
bash tools/pmc_traffic.sh cfg3 512 4 > /dev/null 2>&1; cp gpurun_out/traffic_cfg3.json gpurun_out/r02/traffic_cfg3.json
for f in gpurun_out/r02/bench_*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], '%.3e' % d['value'], '%.3f' % r['frac'], r['kernel'])"; done
