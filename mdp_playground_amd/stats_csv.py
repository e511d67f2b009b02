"""Training / evaluation statistics in the reference's on-disk CSV format, and a loader for it.

The reference writes one "<prefix>.csv" per experiment (mdp_playground/config_processor/
config_processor.py: header `init_stats_file` :241-259, one row per training iteration
`on_train_result` :275-376) and one "<prefix>_eval.csv" (one line per evaluation episode
`on_episode_end` :391-407, "#HACK STRING EVAL" after every training iteration :378-385);
mdp_playground/analysis/analysis.py:15-330 (`MDPP_Analysis.load_data`) is the consumer that pins the
format: space-separated, `#` comment lines, the header's column names split on ", ", the last three
columns timesteps_total / episode_reward_mean / episode_len_mean with timesteps_total restarting at
every run.  Host-side and tiny; the per-episode reductions over batched device tensors are in
EpisodeStats below (one kernel per call on device tensors)."""
from __future__ import annotations

import os

import numpy as np

METRICS = ("timesteps_total", "episode_reward_mean", "episode_len_mean")
EVAL_SEPARATOR = "#HACK STRING EVAL\n"


def format_value(v, config_type="env"):
    """A varied-config value as `on_train_result` writes it, by the type of config it belongs to:
    "env" (:287-301): floats "%.2e", lists "[e1,e2,]" with float elements "%.2e", everything else str() without
    spaces; "agent" (:303-343): floats "%.2e", everything else -- lists included, e.g. fcnet_hiddens "[256,256]" --
    str() without spaces; "model" (:344-349): str() without spaces for every value."""
    if config_type == "model":
        return str(v).replace(" ", "")
    if isinstance(v, float):
        return "%.2e" % v
    if isinstance(v, list) and config_type == "env":
        s = "["
        for e in v:
            s += "%.2e" % e if isinstance(e, float) else str(e)
            s += ","
        return s + "]"
    return str(v).replace(" ", "")


class StatsWriter:
    """Appends to <prefix>.csv / <prefix>_eval.csv what the reference's Ray callbacks append (value formats per
    config type: format_value)."""

    def __init__(self, stats_file_prefix, columns, algorithm, write_header=True, column_types=None):
        """column_types: {column: "env" | "agent" | "model"} -- which config a varied column belongs to (the
        reference formats values per type, format_value); columns not named are env columns."""
        self.prefix, self.columns, self.algorithm = stats_file_prefix, list(columns), str(algorithm)
        self.column_types = dict(column_types or {})
        for c, t in self.column_types.items():
            if t not in ("env", "agent", "model"):
                raise ValueError(f"column_types[{c!r}] must be 'env', 'agent' or 'model'")
        if write_header:                                        # init_stats_file, :241-259
            with open(self.prefix + ".csv", "a") as f:
                f.write("# training_iteration, algorithm, ")
                for c in self.columns:
                    f.write(c + ", ")
                f.write("timesteps_total, episode_reward_mean, episode_len_mean\n")

    def write_train_row(self, training_iteration, config_values, timesteps_total, episode_reward_mean,
                        episode_len_mean, evaluation=False):
        """One training iteration (:275-376).  config_values: {column: value} of the varied configs."""
        with open(self.prefix + ".csv", "a") as f:
            f.write(str(training_iteration) + " " + self.algorithm + " ")
            for c in self.columns:
                f.write(format_value(config_values[c], self.column_types.get(c, "env")) + " ")
            f.write(str(timesteps_total) + " " + "%.2e" % episode_reward_mean + " " + "%.2e" % episode_len_mean + "\n")
        if evaluation:                                          # :378-385
            with open(self.prefix + "_eval.csv", "a") as f:
                f.write(EVAL_SEPARATOR)

    def write_eval_episode(self, episode_reward, episode_length):
        with open(self.prefix + "_eval.csv", "a") as f:         # on_episode_end, :391-407
            f.write("%.2e" % episode_reward + " " + str(episode_length) + "\n")


def _normaliser_episodic_reward(name, dim_val):       # analysis.py:560-567
    if name == "sequence_length":
        return dim_val
    if name == "delay":
        return 100.0 / (100 - dim_val)
    return np.nan


def load_stats(dir_name, exp_name, num_metrics=3, load_eval=False, normalise_episodic_reward=True):
    """The part of MDPP_Analysis.load_data (analysis.py:54-330) that reads the two files back: column
    names, the varied dimensions' values, the last row of every run (timesteps_total restarts), the
    end-of-training stats reshaped to (n_values per varied column ..., num_metrics), the per-run means
    (`train_aucs`), and for the eval file the mean over each iteration's last 10 evaluation episodes."""
    import pandas as pd
    stats_file = os.path.join(dir_name, exp_name)
    stats_pd = pd.read_csv(stats_file + ".csv", skip_blank_lines=True, header=None, comment="#", sep=" ")
    with open(stats_file + ".csv") as f:
        config_names = f.readline().strip().split(", ")
    config_names[0] = config_names[0][2:]
    dims_values, config_counts = [], []
    for i in range(1, len(config_names) - num_metrics):
        dims_values.append(stats_pd[i].unique())
        config_counts.append(stats_pd[i].nunique())
    config_counts.append(num_metrics)
    config_counts = tuple(config_counts)
    final_rows = []
    ts = stats_pd.iloc[:, -num_metrics].to_numpy()
    for i in range(stats_pd.shape[0] - 1):
        if ts[i] > ts[i + 1]:
            final_rows.append(i)
    final_rows.append(stats_pd.shape[0] - 1)
    metrics = stats_pd.iloc[:, -num_metrics:].to_numpy(dtype=float)
    train_stats = np.reshape(metrics[final_rows], config_counts)
    aucs, prev = [], 0
    for fr in final_rows:
        aucs.append(metrics[prev:fr + 1].mean(axis=0))
        prev = fr + 1
    out = {"config_names": config_names[1:], "metric_names": config_names[-num_metrics:], "dims_values": dims_values,
           "config_counts": config_counts, "final_rows": final_rows, "train_stats": train_stats,
           "train_aucs": np.reshape(np.array(aucs), config_counts), "train_curves": stats_pd.to_numpy()}
    out["eval_stats"] = None
    if load_eval:
        eval_stats = np.loadtxt(stats_file + "_eval.csv", dtype=float)
        hack, i = [], 0
        for line in open(stats_file + "_eval.csv"):
            if line.strip().startswith("#HACK"):
                hack.append(i - len(hack))
            i += 1
        ray_0_9_0 = hack[0] == 0
        if ray_0_9_0:
            hack = hack[1:]
        final_10 = [eval_stats[h - 10:h] for h in hack]
        if ray_0_9_0:
            final_10.append(eval_stats[hack[-1]:])
        mean_eval = np.mean(np.array(final_10), axis=1)
        mean_eval = np.concatenate((np.atleast_2d(ts).T, mean_eval), axis=1)
        out["eval_curves"] = mean_eval
        out["eval_stats"] = np.reshape(mean_eval[final_rows, :], config_counts)
    # episodic rewards re-scaled for sequence lengths / delays that were varied (:318-352; the last
    # varied column -- the seeds -- is not a dimension of hardness)
    counts = config_counts[:-1]
    for i in range(len(counts) - 1):
        if counts[i] > 1 and out["config_names"][i] in ("sequence_length", "delay") and normalise_episodic_reward:
            for j in range(counts[i]):
                ind = (slice(None),) * i + (j,) + (slice(None),) * (len(counts) - i - 1) + (1,)
                mult = _normaliser_episodic_reward(out["config_names"][i], dims_values[i][j])
                out["train_stats"][ind] = out["train_stats"][ind] * mult
                out["train_aucs"][ind] = out["train_aucs"][ind] * mult
                if load_eval:
                    out["eval_stats"][ind] *= mult
    return out


class EpisodeStats:
    """episode_reward_mean / episode_len_mean of a batch of env instances from the per-step tensors a
    vector env returns (reward[N] or [K, N], terminated / truncated of the same shape): running per-instance
    return and length, and the sums over the episodes that finished since the last pop() (what RLlib reports
    per training iteration).  Device tensors go through ONE kernel per call (mdpp_episode_stats: a lane per
    instance walks the K rows; a [512, 65 536] rollout is two launches, not 4 000 torch ops); host tensors
    through plain torch ops."""

    def __init__(self, num_envs, device):
        import torch
        self._t = torch
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self.ret = torch.zeros(num_envs, dtype=torch.float64, device=device)
        self.len = torch.zeros(num_envs, dtype=torch.int64, device=device)
        self.sum_ret = torch.zeros((), dtype=torch.float64, device=device)
        self.sum_len = torch.zeros((), dtype=torch.int64, device=device)
        self.count = torch.zeros((), dtype=torch.int64, device=device)
        self.timesteps_total = 0
        self._lib = self._scratch = None
        if self.device.type == "cuda":
            from . import _capi
            self._lib = _capi.load()          # (raises when the HIP library is missing: no fallback for device tensors)
            self._scratch = torch.empty(3 * ((self.num_envs + 255) // 256), dtype=torch.float64, device=device)

    def update(self, reward, ended, ended2=None):
        """reward [N] or [K, N] (float32 / float64); ended (and optionally ended2, e.g. terminated and truncated:
        an episode ends where either is set), same shape."""
        t = self._t
        if reward.dim() == 1:
            reward, ended = reward[None], ended[None]
            ended2 = None if ended2 is None else ended2[None]
        K, N = int(reward.shape[0]), int(reward.shape[1])
        if N != self.num_envs:
            raise ValueError(f"expected {self.num_envs} instances, got {N}")
        if self._lib is not None:
            if reward.dtype not in (t.float32, t.float64):
                reward = reward.to(t.float64)
            r = reward.contiguous()
            e1 = ended.contiguous().view(t.uint8) if ended.dtype == t.bool else ended.to(t.uint8).contiguous()
            e2 = None
            if ended2 is not None:
                e2 = ended2.contiguous().view(t.uint8) if ended2.dtype == t.bool else ended2.to(t.uint8).contiguous()
            if r.device != self.device or e1.device != self.device:
                raise ValueError("EpisodeStats.update: tensors must live on the device the statistics were made for")
            from . import _capi
            with t.cuda.device(self.device):
                rc = self._lib.mdpp_episode_stats(
                    K, N, r.data_ptr(), int(r.dtype == t.float64), e1.data_ptr(), None if e2 is None else e2.data_ptr(),
                    self.ret.data_ptr(), self.len.data_ptr(), self.sum_ret.data_ptr(), self.sum_len.data_ptr(),
                    self.count.data_ptr(), self._scratch.data_ptr(), t.cuda.current_stream(self.device).cuda_stream)
            if rc:
                raise _capi.MdppError(f"mdpp_episode_stats failed ({rc})")
        else:
            for k in range(K):
                self.ret += reward[k].to(t.float64)
                self.len += 1
                e = ended[k].to(t.bool)
                if ended2 is not None:
                    e = e | ended2[k].to(t.bool)
                self.sum_ret += (self.ret * e).sum()
                self.sum_len += (self.len * e).sum()
                self.count += e.sum()
                self.ret.masked_fill_(e, 0.0)
                self.len.masked_fill_(e, 0)
        self.timesteps_total += K * N

    def pop(self):
        """-> (timesteps_total, episode_reward_mean, episode_len_mean) over the episodes finished since the
        last pop(); NaN means when none finished (RLlib's convention)."""
        n = int(self.count.item())
        out = (self.timesteps_total, float(self.sum_ret.item()) / n if n else float("nan"),
               float(self.sum_len.item()) / n if n else float("nan"))
        self.sum_ret.zero_(); self.sum_len.zero_(); self.count.zero_()
        return out
