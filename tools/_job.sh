#!/bin/bash
for w in d_s50_delay4 d_s24_rdist; do python3 tools/ablate.py run mdpp_discrete_quiet.hip $w numpy o0 o6 o0 o6 2>&1 | grep -v "^$" | tail -4; done
