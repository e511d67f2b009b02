#!/bin/bash
for w in cfg2_noise d_s8_rn0; do python3 tools/ablate.py run mdpp_discrete_lean_npnoise.hip $w numpy d ns d ns 2>&1 | grep " us per launch" | cut -c1-40,100-160; done
