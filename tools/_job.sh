#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bench_shape or golden" -p no:cacheprovider 2>&1 | tail -2
