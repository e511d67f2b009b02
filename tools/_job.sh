#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -x -k "sigma_zero or quiet or per_env or random_configurations_spec" -p no:cacheprovider 2>&1 | tail -4
timeout 600 python tools/time_legs.py d_s50_rn0 d_s50_rn0:NO_SIGMA0 d_s24_rdist d_s50_delay4 cfg2_per_env d_s50_rn0 --reps 3 2>&1 | tail -8
