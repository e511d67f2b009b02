#!/usr/bin/env python3
"""CSV-format goldens (second half) — THIS CONTAINER ONLY.

(1) Copies the stats files the reference's own tests hold (tests/files/mdpp_12744267_SAC_target_radius/,
    data, not source) to tests/golden/csv/ and records what the REFERENCE loader
    (mdp_playground/analysis/analysis.py MDPP_Analysis.load_data) makes of them.
(2) Writes a small experiment with mdp_playground_amd.stats_csv.StatsWriter, loads it with the
    reference loader, and records that too: the writer's output is what upstream's analysis code reads.
The reference writer itself (config_processor.py) imports Ray at module level and cannot run here; its
format is restated from its source and pinned by (1) + (2).
"""
import contextlib
import io
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "gymnasium_standin"))
sys.path.insert(1, "/root/reference")
sys.path.insert(2, ROOT)

import numpy as np  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "csv")


def ref_load(dir_name, exp_name, load_eval):
    import matplotlib
    matplotlib.use("Agg")
    with contextlib.redirect_stdout(io.StringIO()):
        from mdp_playground.analysis.analysis import MDPP_Analysis
        an = MDPP_Analysis()
        res = an.load_data(dir_name, exp_name, load_eval=load_eval)
    train_stats, eval_stats, train_curves, eval_curves, train_aucs, eval_aucs = res
    return dict(train_stats=np.asarray(train_stats, float), train_aucs=np.asarray(train_aucs, float),
                eval_stats=np.asarray(eval_stats, float) if load_eval else np.zeros(0),
                eval_curves=np.asarray(eval_curves, float) if load_eval else np.zeros(0),
                final_rows=np.asarray(an.final_rows_for_a_config), config_counts=np.asarray(an.config_counts),
                config_names=np.asarray(an.config_names), metric_names=np.asarray(an.metric_names))


def write_own(dir_name, exp_name):
    from mdp_playground_amd.stats_csv import StatsWriter
    cols = ["delay", "sequence_length", "transition_noise", "target_point", "make_denser", "dummy_seed"]
    w = StatsWriter(os.path.join(dir_name, exp_name), cols, "DQN")
    rng = np.random.default_rng(0)
    for delay in (0, 2):
        for L in (1, 3):
            for seed in (0, 1, 2):
                for it in range(1, 6):
                    w.write_train_row(it, {"delay": delay, "sequence_length": L, "transition_noise": 0.25,
                                           "target_point": [0.0, 1], "make_denser": True, "dummy_seed": seed},
                                      1000 * it, float(rng.normal(10 * it, 1.0)), float(rng.uniform(20, 100)), evaluation=True)
                    for _ in range(10):
                        w.write_eval_episode(float(rng.normal(5 * it, 1.0)), int(rng.integers(10, 100)))


def main():
    os.makedirs(OUT, exist_ok=True)
    src = "/root/reference/tests/files/mdpp_12744267_SAC_target_radius"
    name = "sac_move_to_a_point_target_radius"
    for suf in (".csv", "_eval.csv"):
        shutil.copyfile(os.path.join(src, name + suf), os.path.join(OUT, name + suf))
        os.chmod(os.path.join(OUT, name + suf), 0o644)
    np.savez_compressed(os.path.join(OUT, "ref_loader_on_upstream_files.npz"), **ref_load(OUT, name, True))
    own = "own_writer_dqn"
    for suf in (".csv", "_eval.csv"):
        if os.path.exists(os.path.join(OUT, own + suf)):
            os.remove(os.path.join(OUT, own + suf))
    write_own(OUT, own)
    np.savez_compressed(os.path.join(OUT, "ref_loader_on_own_writer.npz"), **ref_load(OUT, own, True))
    print("ok")


if __name__ == "__main__":
    main()
