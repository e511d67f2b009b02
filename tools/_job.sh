timeout 900 python3 tools/soak_noise.py 6 philox > gpurun_out/soakp.log 2>&1; echo "soak rc=$?" >> gpurun_out/soakp.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "specialised_kernels_equal_general or lean_rollout_kernel or philox" > gpurun_out/t7.log 2>&1
for o in '{}' '{"reward_noise": null}' '{"transition_noise": null}'; do MDPP_RNG=philox timeout 300 python3 tools/time_config.py "$o" 65536 512 20 cfg2_noise 2>/dev/null | tail -1; done
tail -n 7 gpurun_out/soakp.log | cut -c1-220; tail -n 3 gpurun_out/t7.log
