#!/bin/bash
for w in cfg2 cfg2_noise d_s8_rn0 cfg2_irr; do python3 tools/ablate.py run mdpp_discrete_lean.hip $w numpy d ns d ns 2>&1 | grep " us per launch" | cut -c1-150; done
python3 tools/ablate.py run mdpp_discrete_lean.hip cfg2_noise philox d ns 2>&1 | grep " us per launch" | cut -c1-150
for w in d_s50_delay4 d_s50_rn0 cfg2_per_env; do python3 tools/ablate.py run mdpp_discrete_quiet.hip $w numpy d ns d ns 2>&1 | grep " us per launch" | cut -c1-150; done
for w in cfg3 cfg5 c_d2_n0; do python3 tools/ablate.py run mdpp_continuous_fast.hip $w numpy d ns d ns 2>&1 | grep " us per launch" | cut -c1-150; done
python3 tools/ablate.py run mdpp_continuous_fast.hip cfg5 philox d ns 2>&1 | grep " us per launch" | cut -c1-150
