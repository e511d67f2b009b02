#!/bin/bash
# SQ counters of two workloads back to back (GPU box): bash tools/sq_two.sh d_s8_rn0 cfg2
for w in "$@"; do echo "=== $w"; SQ_MORE=1 bash tools/pmc_sq.sh $w 512 2 2>&1 | tail -45; done > gpurun_out/sq_two.txt 2>&1
cat gpurun_out/sq_two.txt
