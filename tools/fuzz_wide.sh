#!/bin/bash
# Exploration runs of the two random-configuration families of tests/test_gpu_sweep.py over other seeds (GPU box):
#   bash tools/fuzz_wide.sh "1,2,3,4" "11,12"     (seeds of the widened family, seeds of the dispatch-condition family)
mkdir -p gpurun_out/fz
export MDPP_FUZZ_WIDE_SEEDS=${1:-606}
[ -n "$2" ] && export MDPP_FUZZ_SEEDS=$2
timeout 2400 python -m pytest tests/test_gpu_sweep.py -m gpu -q -k "${3:-random}" --maxfail 12 -p no:cacheprovider 2>&1 | tail -400 > gpurun_out/fz/wide.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/fz/wide.txt | tail -30
