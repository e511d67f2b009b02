cd $GRAFT_REPO_ROOT
cat > /tmp/t.py <<'PY'
import sys, time; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, bench
wl = bench.WORKLOADS["cfg4"]; N = wl["envs"]
env = bench.make_env(wl, N, torch.device("cuda", 0), "numpy"); env.reset()
acts = bench.make_actions(wl, 64, N, env.device, 1)
out = env.alloc_rollout(64)
for k in range(3): env.rollout(acts, out)
torch.cuda.synchronize()
for rep in range(3):
    env.timer_begin()
    for k in range(5): env.rollout(acts, out)
    ms = env.timer_end(); torch.cuda.synchronize()
print(env.rollout_kernel_name(64), "cfg4 rollout: %.2f us per step" % (ms * 1e3 / 320))
for rep in range(2):
    env.timer_begin()
    for k in range(100): env.step(acts[k % 8])
    ms = env.timer_end(); torch.cuda.synchronize()
print(env.rollout_kernel_name(1), "cfg4 single step %.2f us" % (ms * 1e3 / 100))
PY
python3 /tmp/t.py
MDPP_FORCE_WIDE=1 python3 /tmp/t.py
