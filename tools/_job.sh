timeout 1500 python3 tools/ablate_npnoise.py 0 14 15 16 17 18 19 > gpurun_out/ablate_np5.log 2>&1
grep -v amdgpu.ids gpurun_out/ablate_np5.log
