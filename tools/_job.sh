#!/bin/bash
export MDPP_FUZZ_MORE_SEEDS=61,62,63,64,65,66,67,68
bash tools/fuzz_wide.sh "71,72,73,74,75,76,77,78" "81,82,83,84,85,86" random
