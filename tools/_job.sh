#!/bin/bash
export MDPP_FUZZ_STRIDE=1
export MDPP_FUZZ_MORE_SEEDS=17,111
bash tools/fuzz_wide.sh "606,112" "113" vs_oracle
