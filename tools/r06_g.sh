cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "per_env" 2>&1 | grep -v "^$" | tail -25
python3 - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
for opts in ((), ("NO_TRIO",)):
    wl = bench.WORKLOADS["cfg2_per_env"]
    env = bench.make_env(wl, 65536, dev, "numpy")
    if opts: env.set_kernel_options(*opts)
    acts = bench.action_rotation(wl, 512, 65536, dev, 1)
    out = env.alloc_rollout(512)
    for i in range(3): env.rollout(acts[i % len(acts)], out)
    torch.cuda.synchronize()
    env.timer_begin()
    for i in range(10): env.rollout(acts[i % len(acts)], out)
    ms = env.timer_end()
    print(opts, env.rollout_kernel_name(512), "launch_us", ms * 100, "frac", 18 * 65536 * 512 / (ms * 1e-4) / 8e12)
    env.close()
PY
