for o in '{}' '{"image_ro_quant": 180}' '{"image_ro_quant": 90}'; do timeout 300 python3 tools/time_config.py "$o" 8192 512 6 cfg4 2>/dev/null | tail -1; done
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "image or cfg4" > gpurun_out/t8.log 2>&1; tail -n 3 gpurun_out/t8.log
