#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -x -k "quiet or per_env or every_lane or sigma_zero" -p no:cacheprovider 2>&1 | tail -2
timeout 600 python tools/time_legs.py d_s50_delay4 d_s24_rdist d_s50_rn0 cfg2_per_env --reps 3 2>&1 | tail -4 | cut -c1-40,140-200
