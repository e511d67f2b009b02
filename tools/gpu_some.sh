#!/bin/bash
# bash tools/gpu_some.sh <pytest -k expression> [file]   (GPU box)
mkdir -p gpurun_out/some
timeout 2400 python -m pytest ${2:-tests} -m gpu -q -k "$1" --maxfail 10 -p no:cacheprovider 2>&1 | tail -300 > gpurun_out/some/out.txt
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/some/out.txt | tail -40
