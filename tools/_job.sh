python3 - > gpurun_out/probe.log 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import torch, bench
print(bench.hbm_ceilings(torch.device('cuda:0')))
print(bench.hbm_ceilings(torch.device('cuda:0'), nbytes=4<<30))
PY
grep -v amdgpu gpurun_out/probe.log
