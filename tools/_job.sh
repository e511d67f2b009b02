timeout 1500 python3 tools/ablate_npnoise.py 5 6 7 8 9 10 11 12 13 > gpurun_out/ablate_np3.log 2>&1
grep -v amdgpu.ids gpurun_out/ablate_np3.log
