#!/usr/bin/env python3
"""Which kernel serves which configuration: the library's own answer, mdpp_kernel_name, for the BASELINE
configs of bench.py and for every configuration of the reference's experiment files (tests/golden_sweep/cases.json), as
a shared MDP over 65 536 envs (8 192 with pictures) -- a fused launch of 512 steps (64 with pictures) and a single step.
Writes docs/dispatch.md (markdown).  Needs a GPU (a handle is created per configuration); run through gpurun.

    python3 tools/dispatch_table.py [out.md]
"""
import collections
import os
import sys
import warnings

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
import golden_util as gu  # noqa: E402
from mdp_playground_amd import RLToyVectorEnv  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "docs", "dispatch.md")
dev = torch.device("cuda", 0)

RULES = """## The rules behind the tables (first match wins; sources: the `launch_*` functions named)

| kernel | serves | file: function |
|---|---|---|
| `k_discrete_step_wide<PHILOX,NOISE,UNIT,IRR>` | state_space_size 256 ... 65 535 (16-bit table entries and history fields; the general kernel compiled a second time): every call |
| `k_discrete_step_long<PHILOX,NOISE,UNIT,IRR>` | sequence_length 8 ... 15 (a history of sixteen byte fields; the general kernel compiled a third time): every call |
| `k_discrete_step1` / `k_discrete_step1w` | `mdpp_step` (K = 1): one shared MDP, L <= 3, same-step autoreset or none, no irrelevant sub-space, no episode statistics; S <= 16 without noise (`step1`) or any S <= 255 whose table blob fits the rounds a wave stages -- 8 KiB, 12 KiB with a noise key (`step1w`: unit rewards with delay <= 32, or non-unit sequence rewards without noise) | `mdpp_discrete_step1.hip`: `launch_discrete_step1`; the blob: `mdpp_capi.hip`: `mdpp_upload_discrete_tables` |
| `k_discrete_rollout_lean<...,PHILOX,IRR,NEXT,PN,RN,Z0>` | K >= 32, N >= 256: one shared MDP, unit rewards, L <= 3, S <= 8, A <= 16, delay <= 32, every_n <= 64, max_steps < 65 536; noise on numpy streams only where the S noise categoricals share their thresholds (host check); `Z0`: the reward-noise key with sigma 0 | `mdpp_discrete_lean.hip`: `launch_discrete_lean` (+ `_next`, `_noise`, `_npnoise`) |
| `k_discrete_rollout_pipe` / `_fast` | the same quiet shape up to S = 16 (three roles for long rollouts of full blocks / one role: short rollouts, the state kernel of image handles) | `mdpp_discrete_pipe.hip`, `mdpp_discrete_fast.hip` |
| `k_discrete_rollout_quiet<...,ROLES,PN,RN,PHILOX,NPH,UNIT,SF,PE>` | K >= 16: one shared MDP whose tables fit 60 KiB of LDS, any S <= 255, L <= 7, irrelevant sub-space, both noises, non-unit sequence rewards (`UNIT=0`: numpy streams, no irrelevant sub-space); ROLES = 2 / 3 for full blocks and K >= 32 (3: autoreset without reward noise = a start-state queue wave; or reward noise alone on numpy streams = `XR`, the env stream by position); `SF`: ROLES = 3, L = 1, same-step autoreset, no step limit, every-step pay, S <= 128; `PE`: one MDP per env, S <= 16, unit rewards, numpy streams, no transition noise | `mdpp_discrete_quiet.hip`: `launch_discrete_quiet`, `launch_discrete_quiet_nu` |
| `k_discrete_step<PHILOX,NOISE,UNIT,LDSTAB,IRR>` | everything else (per-env MDPs in short rollouts, custom R(s, a), tables beyond LDS, episode statistics, ...) | `mdpp_discrete.hip`: `launch_discrete_step` |
| `k_continuous_step1<...,PAR>` | `mdpp_step` on the fast continuous shapes (`PAR = 1`: transition noise on numpy streams at D >= 8) | `mdpp_continuous_fast.hip`: `launch_continuous_step1` |
| `k_continuous_rollout_fast<D,ORDER,NREL,NOISE,HELPER,GEN,PHILOX,NPROD,Z0>` | move_to_a_point, relevant dimensions = the first n_rel, no pictures, an explicit target_point, <= 8 terminal cubes, (D, order, n_rel) one of the built shapes (D in {2, 4, 8, 12}, orders 1-2, n_rel = D or a prefix; D = 2 also order 3); next-step autoreset only without noise or on Philox streams; `HELPER`: noise, K >= 16, full blocks; `NPROD = 2` on numpy streams (generator / walker / consumer) at D >= 8 or D = 2; `Z0`: D = 2 with every present noise key at sigma 0 | `mdpp_continuous_fast.hip`: `launch_continuous_fast`; the shape flag: `mdpp_capi.hip` (`fast_ok`, `next_ok`) |
| `k_continuous_line_rollout` | move_along_a_line with every dimension relevant, D in {2, 4}, order <= 2, no noise, delay 0, K >= 4 | `mdpp_continuous_line.hip`: `launch_continuous_line` |
| `k_continuous_step<DMAX,OMAX,PHILOX,NL>` | everything else (`NL = 8`: the line fit with 5-8 relevant of <= 12 dimensions; beyond that the fit's matrices live in an HBM workspace) | `mdpp_continuous.hip`: `launch_continuous_step` |
| `k_image_step1<NST,PHILOX,WIDE>` | `mdpp_step` on polygon-picture handles the fast / wide renderer serves | `mdpp_image.hip` |
| `k_image_obs_fast<NST>` / `k_image_obs_wide` / `k_image_obs` | polygon pictures: padded templates up to 64 / 128 pixels wide with the polygon never leaving the picture (host check) / the general renderer | `mdpp_image.hip`: `launch_image_obs`; `mdpp_capi.hip`: `mdpp_upload_image_templates` (`img_fast_ok`, `img_colb`) |
| `k_grid_rollout_fast` / `k_grid_step`, `k_imagec_obs` | grid envs; pictures of continuous and grid envs | `mdpp_grid.hip`, `mdpp_imagec.hip` |

`mdpp_set_options` (`MDPP_OPT_NO_*`, `include/mdpp.h`) takes kernels out of this order per handle; `mdpp_kernel_name` answers for a handle and a launch length.

"""
rows = []


def shape_of(cfg):
    if cfg.get("state_space_type") == "discrete":
        s = f"S={cfg.get('state_space_size')} A={cfg.get('action_space_size')} L={cfg.get('sequence_length', 1)} delay={cfg.get('delay', 0)}"
        if cfg.get("image_representations"):
            s += f" image {cfg.get('image_width', 100)}x{cfg.get('image_height', 100)} [{cfg.get('image_transforms', 'none')}]"
    elif cfg.get("state_space_type") == "grid":
        s = f"grid {cfg.get('grid_shape')}"
    else:
        s = f"D={cfg.get('state_space_dim')} order={cfg.get('transition_dynamics_order', 1)} {cfg.get('reward_function', 'move_to_a_point')}"
        if cfg.get("image_representations"):
            s += " image"
    for k in ("transition_noise", "reward_noise"):
        if k in cfg:
            s += f" {k}={cfg[k]}"
    for k in ("reward_every_n_steps", "reward_dist", "irrelevant_features", "diameter"):
        if cfg.get(k) not in (None, False, 1):
            s += f" {k}={cfg[k]}"
    return s


def names(cfg, rng):
    image = bool(cfg.get("image_representations"))
    N, F = (8192, 64) if image else (65536, 512)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env = RLToyVectorEnv(num_envs=N, device=dev, autoreset="same_step", rng=rng, **cfg)
    a, b = env.rollout_kernel_name(F), env.rollout_kernel_name(1)
    env.close()
    return a, b


for wname in ("cfg2", "cfg2_noise", "cfg3", "cfg4", "cfg5", "img100_all", "img100_shift", "d_s50_delay4", "d_s24_rdist", "d_s8_rn0",
              "d_s50_rn0", "c_d2_n0", "grid", "img_cont", "line", "cfg2_irr"):
    cfg = bench.WORKLOADS[wname]["config"]
    for rng in ("numpy", "philox"):
        try:
            a, b = names(cfg, rng)
        except Exception as e:            # (a configuration the library refuses: say so)
            a = b = f"refused: {type(e).__name__}"
        rows.append((f"bench `{wname}`", rng, shape_of(cfg), a, b, 1))
sweep = sorted(k for k in gu.CASES if "_x" in k and k.rsplit("_x", 1)[-1].isdigit())
groups = collections.OrderedDict()
for name in sweep:
    cfg = gu.case_config(name)
    try:
        a, b = names(cfg, "numpy")
    except Exception as e:
        a = b = f"refused: {type(e).__name__}"
    key = (a, b)
    groups.setdefault(key, []).append((name, cfg))
with open(out, "w") as f:
    f.write("# Dispatch: which kernel serves which configuration\n\n"
            "Generated by `tools/dispatch_table.py` on an MI355X from the library's own answer (`mdpp_kernel_name`): a shared MDP over 65 536 env "
            "instances (8 192 with pictures), same-step autoreset; `rollout` = a fused launch of 512 steps (64 with pictures), `step` = "
            "`mdpp_step`.  Template arguments are the dispatch's run-time decisions; every specialised kernel is checked against the "
            "general one on every env by `tests/test_gpu_sweep.py`, the general ones against the reference-generated goldens.\n\n"
            + RULES +
            "## BASELINE / bench workloads\n\n| workload | rng | shape | rollout kernel | step kernel |\n|---|---|---|---|---|\n")
    for w, rng, shp, a, b, _ in rows:
        f.write(f"| {w} | {rng} | {shp} | `{a}` | `{b}` |\n")
    f.write(f"\n## The reference's experiment files ({len(sweep)} unique env configurations, numpy streams), grouped by the kernels chosen\n\n"
            "| configurations | example shape | rollout kernel | step kernel |\n|---|---|---|---|\n")
    for (a, b), members in groups.items():
        ids = ", ".join(n for n, _ in members[:6]) + (f", ... ({len(members)} in all)" if len(members) > 6 else "")
        f.write(f"| {ids} | {shape_of(members[0][1])} | `{a}` | `{b}` |\n")
print("wrote", out, len(rows), "bench rows,", len(groups), "sweep groups")
