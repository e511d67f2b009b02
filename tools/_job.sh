timeout 1500 python3 tools/ablate_walk.py 14 15 16 17 18 19 20 21 > gpurun_out/ablate_walk5.log 2>&1
grep -v amdgpu.ids gpurun_out/ablate_walk5.log
