#!/bin/bash
export MDPP_FUZZ_MORE_SEEDS=91,92,93
bash tools/fuzz_wide.sh "94,95,96,97" "98,99,100" vs_oracle
