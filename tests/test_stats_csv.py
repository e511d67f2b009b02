"""The on-disk statistics format: mdp_playground_amd/stats_csv.py against
(1) the stats files the reference's own tests hold and what the REFERENCE loader
(mdp_playground/analysis/analysis.py:15-330) made of them in the build container
(tests/golden/csv/ref_loader_on_upstream_files.npz, tools/refgen/gen_golden_csv.py), and
(2) a file written by StatsWriter that the reference loader read back (ref_loader_on_own_writer.npz)."""
import filecmp
import os
import sys

import numpy as np
import torch

import golden_util as gu
from mdp_playground_amd import stats_csv

CSV = os.path.join(gu.GOLDEN, "csv")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "refgen"))


def _check(mine, ref, with_eval=True):
    assert list(mine["config_names"]) == [str(x) for x in ref["config_names"]]
    assert list(mine["metric_names"]) == [str(x) for x in ref["metric_names"]]
    assert list(mine["final_rows"]) == list(ref["final_rows"])
    assert tuple(mine["config_counts"][:-1]) == tuple(ref["config_counts"])
    assert np.array_equal(mine["train_stats"], ref["train_stats"])
    assert np.allclose(mine["train_aucs"], ref["train_aucs"], rtol=1e-12, atol=0)
    if with_eval:
        assert np.allclose(mine["eval_stats"], ref["eval_stats"], rtol=1e-12, atol=0)
        assert np.allclose(mine["eval_curves"], ref["eval_curves"], rtol=1e-12, atol=0)


def test_loader_restatement_equals_reference_loader_on_upstream_files():
    ref = np.load(os.path.join(CSV, "ref_loader_on_upstream_files.npz"))
    mine = stats_csv.load_stats(CSV, "sac_move_to_a_point_target_radius", load_eval=True)
    _check(mine, ref)
    assert mine["train_stats"].shape[-1] == 3 and mine["train_stats"].size > 30


def test_writer_output_is_what_the_reference_loader_read(tmp_path):
    import gen_golden_csv                      # (only its deterministic write_own(); nothing of the reference is imported)
    gen_golden_csv.write_own(str(tmp_path), "own_writer_dqn")
    for suf in (".csv", "_eval.csv"):          # byte-identical to the files the reference loader was run on
        assert filecmp.cmp(os.path.join(tmp_path, "own_writer_dqn" + suf), os.path.join(CSV, "own_writer_dqn" + suf), shallow=False)
    ref = np.load(os.path.join(CSV, "ref_loader_on_own_writer.npz"))
    mine = stats_csv.load_stats(str(tmp_path), "own_writer_dqn", load_eval=True)
    _check(mine, ref)
    assert mine["train_stats"].shape == (1, 2, 2, 1, 1, 1, 3, 3)     # algorithm, delay, seq_len, noise, target, denser, seeds, metrics
    with open(os.path.join(tmp_path, "own_writer_dqn.csv")) as f:
        assert f.readline() == ("# training_iteration, algorithm, delay, sequence_length, transition_noise, target_point, "
                                "make_denser, dummy_seed, timesteps_total, episode_reward_mean, episode_len_mean\n")


def test_value_formatting_follows_on_train_result():
    assert stats_csv.format_value(0.05) == "5.00e-02" and stats_csv.format_value(3) == "3"
    assert stats_csv.format_value([0.0, 1, 2.5]) == "[0.00e+00,1,2.50e+00,]"
    assert stats_csv.format_value((0.8, 1.25)) == "(0.8,1.25)" and stats_csv.format_value(True) == "True"


def test_episode_stats_reductions():
    es = stats_csv.EpisodeStats(3, "cpu")
    rew = torch.tensor([[1.0, 2.0, 0.5], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5], [1.0, 2.0, 0.5]])
    end = torch.tensor([[0, 0, 1], [1, 0, 0], [0, 0, 1], [1, 1, 0]], dtype=torch.bool)
    es.update(rew, end)
    ts, rmean, lmean = es.pop()
    # finished episodes: env2 (0.5, len 1), env0 (2.0, len 2), env2 (1.0, len 2), env0 (2.0, len 2), env1 (8.0, len 4)
    assert ts == 12 and abs(rmean - (0.5 + 2.0 + 1.0 + 2.0 + 8.0) / 5) < 1e-12 and abs(lmean - 11 / 5) < 1e-12
    assert np.isnan(es.pop()[1])


def test_value_formats_per_config_type(tmp_path):
    """the reference formats a varied value by the config it belongs to (config_processor.py:287-349):
    env lists "[a,b,]" with "%.2e" floats; agent lists and everything non-float str() without spaces; model values
    str() without spaces even when they are floats."""
    assert stats_csv.format_value([0.25, 3], "env") == "[2.50e-01,3,]"
    assert stats_csv.format_value([256, 256], "agent") == "[256,256]"
    assert stats_csv.format_value(0.001, "agent") == "1.00e-03"
    assert stats_csv.format_value([[16, [8, 8], 4], [32, [4, 4], 2]], "model") == "[[16,[8,8],4],[32,[4,4],2]]"
    assert stats_csv.format_value(0.5, "model") == "0.5"
    prefix = str(tmp_path / "run")
    w = stats_csv.StatsWriter(prefix, ["delay", "fcnet_hiddens", "conv_filters"], "DQN",
                              column_types={"fcnet_hiddens": "agent", "conv_filters": "model"})
    w.write_train_row(1, {"delay": 2, "fcnet_hiddens": [256, 256], "conv_filters": [[16, [8, 8], 4]]}, 1000, 1.5, 20.0)
    lines = open(prefix + ".csv").read().splitlines()
    assert lines[0] == "# training_iteration, algorithm, delay, fcnet_hiddens, conv_filters, timesteps_total, episode_reward_mean, episode_len_mean"
    assert lines[1] == "1 DQN 2 [256,256] [[16,[8,8],4]] 1000 1.50e+00 2.00e+01"
