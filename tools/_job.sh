cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q -x -k "image or img or polygon or step1" 2>&1 | tail -8
python3 - <<'PY'
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch, bench
wl = bench.WORKLOADS["img100_all"]; N = wl["envs"]
for opt in ((), ("NO_STEP1",), ("NO_IMGFAST",)):
    env = bench.make_env(wl, N, torch.device("cuda", 0), "numpy")
    if opt: env.set_kernel_options(*opt)
    env.reset()
    acts = bench.make_actions(wl, 32, N, env.device, 1)
    for _ in range(5): env.step(acts[0])
    torch.cuda.synchronize()
    for rep in range(2):
        env.timer_begin()
        for k in range(100): env.step(acts[k % 8])
        ms = env.timer_end(); torch.cuda.synchronize()
    print("img100_all %s %s: single step %.2f us" % (opt, env.rollout_kernel_name(1), ms * 1e3 / 100), flush=True)
    env.close()
PY
