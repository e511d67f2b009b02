#!/bin/bash
timeout 600 python bench.py --workload img_cont --no-cpu-baseline --no-pmc --no-workloads --no-single-step --detail-out gpurun_out/imgc.json 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['launch_us'], d['roofline']['kernel'])"
