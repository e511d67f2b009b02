#!/bin/bash
# The specialised-equals-general random families at bench-sized env counts (GPU box): bash tools/fuzz_big.sh "65536,16384,40000"
mkdir -p gpurun_out/fz
export MDPP_FUZZ_SIZES=${1:-65536}
export MDPP_FUZZ_WIDE_SEEDS=${2:-606}
timeout 2400 python -m pytest tests/test_gpu_sweep.py -m gpu -q -k "specialised_equals_general and random" --maxfail 12 -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/fz/big.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/fz/big.txt | tail -30
