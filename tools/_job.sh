#!/bin/bash
timeout 300 python tools/repro_c14.py 2>&1 | grep "^philox" | cut -c1-160
export MDPP_FUZZ_MORE_SEEDS=101,102,103
bash tools/fuzz_wide.sh "104,105,106,107" "108" vs_oracle
