#!/bin/bash
# Round-2 HEAD (build/r2tree, extracted + built by hand: git archive c3b1936 mdp_playground_amd include | tar -x -C build/r2tree)
# against this tree on ONE lease, alternating processes.  usage: bash tools/bisect_lean.sh [rounds]
R=${1:-3}
mkdir -p gpurun_out
: > gpurun_out/bisect_lean.jsonl
for r in $(seq 1 $R); do
  for t in build/r2tree .; do
    timeout 300 python3 tools/bisect_lean.py $t 5 20 512 numpy | tail -1 >> gpurun_out/bisect_lean.jsonl
  done
done
for t in build/r2tree .; do
  timeout 300 python3 tools/bisect_lean.py $t 5 20 512 philox | tail -1 >> gpurun_out/bisect_lean.jsonl
done
cat gpurun_out/bisect_lean.jsonl
