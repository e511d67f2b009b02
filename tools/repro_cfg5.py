import sys, torch, numpy as np
sys.path.insert(0, ".")
from mdp_playground_amd import RLToyVectorEnv
dev = torch.device("cuda", 0)
ccfg = dict(state_space_type="continuous", state_space_dim=12, relevant_indices=[0, 1, 2, 3], irrelevant_features=True,
            target_point=[0, 0, 0, 0], target_radius=0.05, state_space_max=10, action_space_max=1,
            transition_dynamics_order=2, inertia=1, time_unit=0.1, transition_noise=0.05, reward_noise=0.05,
            make_denser=True, reward_function="move_to_a_point", seed=0)
for N in (4096, 65536):
  for presteps in (0, 24):
    for trial in range(3):
        a = RLToyVectorEnv(num_envs=N, device=dev, autoreset="same_step", **ccfg)
        b = RLToyVectorEnv(num_envs=N, device=dev, autoreset="same_step", **ccfg)
        g = torch.Generator(device=dev); g.manual_seed(5)
        ok = True
        for t in range(presteps):
            x = torch.rand((N, 12), generator=g, device=dev) * 2 - 1
            ra, rb = a.step(x), b.step(x)
            ok = ok and all(torch.equal(p, q) for p, q in zip(ra[:4], rb[:4]))
        acts = torch.rand((64, N, 12), generator=g, device=dev) * 2 - 1
        ra, rb = a.rollout(acts), b.rollout(acts)
        eq = [torch.equal(p, q) for p, q in zip(ra, rb)]
        nd = int((ra[0] != rb[0]).any(dim=2).any(dim=0).sum()) if not eq[0] else 0
        first = int((ra[0] != rb[0]).any(dim=2).any(dim=1).nonzero()[0]) if not eq[0] else -1
        sa, sb = a.get_rng_streams(0), b.get_rng_streams(0)
        print(N, presteps, trial, "steps_ok", ok, "rollout eq", eq, "envs differing", nd, "first step", first, "streams eq", np.array_equal(sa, sb), a.rollout_kernel_name(64), "status", int(a.status().any()), int(b.status().any()), flush=True)
        a.close(); b.close()
