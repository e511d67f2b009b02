#!/bin/bash
export MDPP_FUZZ_MORE_SEEDS=1,2,3,4,5,6,7,8,9,10,11,12
bash tools/fuzz_wide.sh "606" "" random
