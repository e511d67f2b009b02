#!/usr/bin/env python3
"""Throughput of the batched post-processor (mdpp_post_step_n) on the GPU box:
fused K-step calls on synthetic inner-env outputs resident in HBM, HIP-event time per call, algorithmic
bytes per instance-step against the 8 TB/s HBM peak.   python3 tools/bench_post.py [workload ...]"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mdp_playground_amd.post import VectorPostProcessor  # noqa: E402

WORKLOADS = {
    # reward in 8 + done 1 + reward out 8 (the FIFO of delayed rewards lives in registers / LDS for the call since
    # round 2: its slots are not HBM traffic any more; round-2 lines before that counted 33 B)
    "post_disc": dict(N=65536, K=256, alg=17, args=dict(n_actions=8),
                      cfg=dict(state_space_type="discrete", delay=4, reward_scale=2.0, reward_shift=-1.0, seed=0)),
    "post_disc_noise": dict(N=65536, K=256, alg=17, args=dict(n_actions=8),
                            cfg=dict(state_space_type="discrete", delay=4, reward_noise=0.1, reward_scale=2.0, seed=0)),
    # + observations float32[12] in and out
    "post_cont_noise": dict(N=65536, K=128, alg=17 + 96, args=dict(obs_shape=(12,), obs_dtype=np.float32),
                            cfg=dict(state_space_type="continuous", delay=4, transition_noise=0.05, reward_noise=0.05, seed=0)),
    # Atari-sized frames: 84 x 84 x 3 in, 124 x 124 x 3 out (image_padding 20), shift drawn per frame
    "post_img": dict(N=4096, K=8, alg=17 + 84 * 84 * 3 + 124 * 124 * 3, args=dict(n_actions=6, obs_shape=(84, 84, 3)),
                     cfg=dict(state_space_type="discrete", image_transforms="shift", image_padding=20, image_sh_quant=1, seed=0)),
}


def run(name, rng="numpy", reps=10):
    w = WORKLOADS[name]
    N, K = w["N"], w["K"]
    post = VectorPostProcessor(N, rng=rng, autoreset=True, **w["args"], **w["cfg"])
    dev = post.device
    g = torch.Generator(device=dev); g.manual_seed(1)
    rew = torch.randint(-8, 9, (K, N), generator=g, device=dev).double() / 4
    done = torch.rand((K, N), generator=g, device=dev) < 0.02
    obs = out = None
    if "obs_shape" in w["args"]:
        shp = (K, N) + tuple(w["args"]["obs_shape"])
        if post.image:
            obs = torch.randint(0, 256, shp, generator=g, device=dev, dtype=torch.uint8)
            post.reset(obs[0])
            out = torch.empty((K, N) + post._obs_out_shape, dtype=torch.uint8, device=dev)
        else:
            obs = torch.randn(shp, generator=g, device=dev)
            out = torch.empty_like(obs)
            post.reset()
    else:
        post.reset()
    for _ in range(2):
        post.step(obs, rew, done, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        post.step(obs, rew, done, out=out)
    e1.record(); e1.synchronize()
    per_call = e0.elapsed_time(e1) * 1e-3 / reps
    rate = N * K / per_call
    gbs = rate * w["alg"] / 1e9
    line = {"workload": name, "rng": rng, "instances": N, "steps_per_call": K, "instance_steps_per_s": rate,
            "call_us": per_call * 1e6, "alg_bytes_per_instance_step": w["alg"],
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0}}
    post.close()
    return line


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in WORKLOADS] or list(WORKLOADS)
    for n in names:
        for rng in ("numpy", "philox"):
            print(json.dumps(run(n, rng)), flush=True)
