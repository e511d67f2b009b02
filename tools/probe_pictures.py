"""The picture renderer's store shape without anything else (mdpp_probe_hbm modes 5 / 6) beside plain fills, GB/s.
    python3 tools/probe_pictures.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdp_playground_amd import _capi  # noqa: E402

lib = _capi.load()
dev = torch.device("cuda", 0)
nbytes = 7056 * 4 * 32768 * 4          # 3.7 GB: one 64-step batch of cfg4
dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for name, mode in (("fill 16 KiB tiles", 1), ("fill 16 KiB tiles nt", 4), ("pictures, one per wave", 5), ("pictures, workgroup region", 6),
                   ("pictures, workgroup region unaligned", 7), ("pictures, one per wave", 5), ("pictures, workgroup region unaligned", 7),
                   ("pictures, workgroup region", 6), ("pictures, 8 per workgroup, 2 per wave", 8), ("pictures, one per wave", 5),
                   ("pictures, 8 per workgroup, 2 per wave", 8), ("fill 16 KiB tiles", 1)):
    ms = ctypes.c_float(0.0)
    rc = lib.mdpp_probe_hbm(mode, ctypes.c_void_p(dst.data_ptr()), None, nbytes, 5, stream, ctypes.byref(ms))
    print(f"{name:32s} rc {rc}  {nbytes * 5 / (ms.value * 1e-3) / 1e9:8.1f} GB/s", flush=True)
t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t.fill_(1); e0.record()
for _ in range(5):
    t.fill_(2)
e1.record(); torch.cuda.synchronize()
print(f"{'torch fill_':32s}       {nbytes * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9:8.1f} GB/s")
