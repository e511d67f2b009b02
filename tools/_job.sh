timeout 900 python3 tools/soak_cfg5.py 6 > gpurun_out/soak5.log 2>&1; echo "soak rc=$?" >> gpurun_out/soak5.log
for i in 1 2; do timeout 300 python3 tools/time_config.py '{}' 65536 512 10 cfg5 >> gpurun_out/time5.log 2>&1; done
timeout 300 python3 tools/time_config.py '{"reward_noise": null}' 65536 512 10 cfg5 >> gpurun_out/time5.log 2>&1
timeout 600 python3 -m pytest tests/test_integration_stub.py tests/test_gpu_boundary.py -x -q -m gpu -k "stub or line_reward" > gpurun_out/t1.log 2>&1
tail -n 8 gpurun_out/soak5.log; tail -n 8 gpurun_out/time5.log; tail -n 5 gpurun_out/t1.log
