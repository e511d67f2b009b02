cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q -x -k "image or img or polygon or step1 or graph" 2>&1 | tail -6
python3 - <<'PY'
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
from mdp_playground_amd import RLToyVectorEnv
dev = torch.device("cuda", 0)
cfg = dict(state_space_type="discrete", action_space_type="discrete", state_space_size=[8, 11], action_space_size=[8, 11],
                       irrelevant_features=True, delay=0, image_representations=True, image_width=84, image_height=84,
                       image_transforms="shift,rotate", seed=2)
N = 8192
for opt in ((), ("NO_STEP1",)):
    env = RLToyVectorEnv(num_envs=N, device=dev, autoreset="same_step", **cfg)
    if opt: env.set_kernel_options(*opt)
    env.reset()
    rng = np.random.default_rng(0)
    a = np.stack([rng.integers(0, x, size=(8, N)) for x in (8, 11)], axis=-1).astype(np.int32)
    acts = torch.as_tensor(a, device=dev)
    for _ in range(20): env.step(acts[0])
    torch.cuda.synchronize(); best = 1e9
    for rep in range(3):
        env.timer_begin()
        for k in range(200): env.step(acts[k % 8])
        ms = env.timer_end(); torch.cuda.synchronize(); best = min(best, ms * 1e3 / 200)
    print("irr84 %s %s: %.2f us per step" % (opt, env.rollout_kernel_name(1), best), flush=True)
    env.close()
PY
