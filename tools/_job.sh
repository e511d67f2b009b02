#!/bin/bash
mkdir -p gpurun_out/some
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -x -k "sigma_zero or noise or bench_shape or lean" -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/time_legs.py d_s8_rn0 d_s8_rn0:NO_SIGMA0 cfg2_noise cfg2 d_s8_rn0 cfg2_noise --reps 3 2>&1 | tail -12
