#!/bin/bash
export MDPP_FUZZ_MORE_SEEDS=17,1,2,3,4,5
bash tools/fuzz_wide.sh "606" "" random
