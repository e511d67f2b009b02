timeout 300 python3 tools/time_config.py '{}' 65536 512 10 cfg2_noise > gpurun_out/timen.log 2>&1
timeout 300 python3 tools/time_config.py '{"reward_noise": null}' 65536 512 10 cfg2_noise >> gpurun_out/timen.log 2>&1
timeout 300 python3 tools/time_config.py '{"transition_noise": null}' 65536 512 10 cfg2_noise >> gpurun_out/timen.log 2>&1
timeout 900 python3 tools/soak_noise.py 6 numpy > gpurun_out/soakn.log 2>&1; echo "soak rc=$?" >> gpurun_out/soakn.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "discrete_shared_mdp_4096 or specialised_kernels_equal_general or discrete_fused_rollout_vs_reference_golden" > gpurun_out/t4.log 2>&1
grep -v amdgpu.ids gpurun_out/timen.log; tail -n 12 gpurun_out/soakn.log; tail -n 8 gpurun_out/t4.log
