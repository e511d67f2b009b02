#!/usr/bin/env python3
"""Golden vectors for the env configurations the reference's OWN experiment sweeps use.

THIS CONTAINER ONLY (reads /root/reference/experiments/*.py).  Every experiment file whose env is "RLToy-v0" is executed
with a stand-in for `ray.tune` (the files only build dicts); its static `env_config["env_config"]` is merged with a "star"
over `var_env_configs` (every value of every swept key once, the other keys at their first value; `dummy_seed` dropped,
`log_filename` dropped: a time-stamped path).  Duplicates across files are merged.  Each unique config is then run through
tools/refgen/gen_golden.py's recorder (same .npz layout as every other golden, E = 1, T steps, reset on done).

    python tools/refgen/gen_sweep.py list                 # the unique configs and the experiments that use them
    python tools/refgen/gen_sweep.py <outdir> [T] [names] # record them (all, or the named ones) into <outdir> + cases.json
"""
import glob
import json
import os
import sys
import types
from collections import OrderedDict

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def experiment_configs():
    ray, tune = types.ModuleType("ray"), types.ModuleType("ray.tune")
    tune.grid_search = lambda v: {"grid_search": v}
    tune.__getattr__ = lambda name: (lambda *a, **k: None)
    ray.tune = tune
    sys.modules["ray"], sys.modules["ray.tune"] = ray, tune
    uniq = OrderedDict()
    cwd = os.getcwd()
    os.chdir("/root/reference")                   # (some files open paths relative to the repository root)
    try:
        for f in sorted(glob.glob("/root/reference/experiments/*.py")):
            g = {"__name__": "exp", "__file__": f}
            try:
                with open(os.devnull, "w") as dn:
                    so, sys.stdout = sys.stdout, dn
                    try:
                        exec(compile(open(f).read(), f, "exec"), g)
                    finally:
                        sys.stdout = so
            except Exception:
                continue
            ec = g.get("env_config", {})
            if ec.get("env") != "RLToy-v0" or not ec.get("env_config"):
                continue
            static = dict(ec["env_config"])
            var = g.get("var_env_configs") or (g.get("var_configs") or {}).get("env") or OrderedDict()
            base = {k: v[0] for k, v in var.items() if k != "dummy_seed"}
            star = [dict(base)]
            for k, vals in var.items():
                if k != "dummy_seed":
                    star += [dict(base, **{k: v}) for v in vals[1:]]
            for c in star:
                m = dict(static, **c)
                m.pop("log_filename", None)
                key = json.dumps(m, sort_keys=True, default=list)
                uniq.setdefault(key, []).append(os.path.basename(f)[:-3])
    finally:
        os.chdir(cwd)
    out = []
    for key, exps in uniq.items():
        cfg = json.loads(key)
        kind = cfg.get("state_space_type")
        pre = "i" if cfg.get("image_representations") else ("d_irr" if isinstance(cfg.get("state_space_size"), list) else kind[0])
        out.append((f"{pre}_x{len(out):03d}", cfg, sorted(set(exps))))
    return out


def main():
    cases = experiment_configs()
    if sys.argv[1] == "list":
        for name, cfg, exps in cases:
            print(name, json.dumps(cfg, sort_keys=True), exps[:3], len(exps))
        print(len(cases), "unique configs")
        return
    out = os.path.abspath(sys.argv[1])
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    only = set(sys.argv[3:])
    import gen_golden as G
    os.makedirs(out, exist_ok=True)
    G.OUT = out
    meta = {}
    for name, cfg, exps in cases:
        if only and name not in only:
            continue
        cfg = dict(cfg)
        seed = cfg.pop("seed", None)
        if "image_scale_range" in cfg:
            cfg["image_scale_range"] = tuple(cfg["image_scale_range"])
        case = dict(config=cfg, seeds=[seed], T=T, reset="on_done")
        try:
            G.run_case(name, case)
        except Exception as e:
            print(f"{name}: REFERENCE RAISED {type(e).__name__}: {str(e)[:200]}")
            continue
        meta[name] = G.jsonable(dict(case, experiments=exps))
    with open(os.path.join(out, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
