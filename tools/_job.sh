timeout 1500 python3 -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/t6.log 2>&1
timeout 900 python3 tools/soak_cfg5.py 3 > gpurun_out/soak5.log 2>&1; echo "soak rc=$?" >> gpurun_out/soak5.log
tail -n 5 gpurun_out/t6.log; tail -n 3 gpurun_out/soak5.log
