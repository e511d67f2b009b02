#!/bin/bash
export MDPP_FUZZ_STRIDE=1
export MDPP_FUZZ_MORE_SEEDS=121
bash tools/fuzz_wide.sh "122,123" "124,125" "random and vs_oracle"
