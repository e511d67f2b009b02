#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s); python bench.py > gpurun_out/bench_default.txt 2>/dev/null; echo "bench.py (no flags): rc $? wall $(( $(date +%s) - t0 )) s, last line $(tail -n 1 gpurun_out/bench_default.txt | wc -c) bytes"
tail -n 1 gpurun_out/bench_default.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['steps'], d['warmup'])"
