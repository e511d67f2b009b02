for r in 1 2; do timeout 1200 python3 tools/ablate.py run mdpp_discrete_lean.hip cfg2 numpy base limbs limbs_h2 limbs_h1 limbs_hmin32 limbs_hmin8; done > gpurun_out/ablate_lean4.log 2>&1
timeout 1800 python3 -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "step_graph" > gpurun_out/t5.log 2>&1
grep -v amdgpu.ids gpurun_out/ablate_lean4.log | cut -c1-12,95-; tail -n 3 gpurun_out/t5.log
