#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -x -k "image or i_100 or pictures or wide" -p no:cacheprovider 2>&1 | tail -2
python3 tools/ablate.py run mdpp_image.hip img100_all numpy c0 c1 c0 c1 2>&1 | grep -v "^$" | tail -4
bash tools/pmc_traffic.sh img100_all 64 3 2>&1 | tail -6
