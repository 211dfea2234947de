"""Minimal experiment runner for the latent-factor plugins (single configuration, no hyperopt).

Honours the slice of Elliot's YAML schema (elliot/namespace/namespace_model.py:28-61) that the hello-world style
experiments use: dataset, data_config {strategy: dataset|fixed, dataset_path | train_path/test_path/validation_path}, prefiltering, binarize,
splitting {test_splitting, validation_splitting: every strategy of base_splitter.py}, negative_sampling {strategy: random|fixed}, top_k, evaluation {cutoffs, simple_metrics,
relevance_threshold}, gpu, path_output_rec_*, models {<Model>: {meta: {...}, <hyper-params>}}.
The full driver (HPO, result handlers, statistical tests: elliot/run.py:39-148) stays Elliot's: plug the models in
there through elliot_amd/external/__init__.py.
"""
import logging
import os
import sys
from types import SimpleNamespace

import numpy as np
import yaml

from . import recommender as rec


def _ns(d):
    return SimpleNamespace(**{k: v for k, v in d.items()})


def _resolve(base_dir, path, dataset):
    path = path.format(dataset)
    return path if os.path.isabs(path) else os.path.abspath(os.path.join(base_dir, path))


def build_config(exp, base_dir):
    ds = exp.get("dataset", "dataset")
    ev = dict(exp.get("evaluation", {}))
    ev.setdefault("simple_metrics", ["nDCG"])
    ev.setdefault("relevance_threshold", 0)             # run.py:55-56
    ev.setdefault("paired_ttest", False)
    ev.setdefault("wilcoxon_test", False)
    ev.setdefault("complex_metrics", [])
    cfg = SimpleNamespace(
        dataset=ds, top_k=exp.get("top_k", 10), evaluation=_ns(ev), config_test=False, gpu=exp.get("gpu", 0),
        path_output_rec_result=_resolve(base_dir, exp.get("path_output_rec_result", "../results/{0}/recs/"), ds),
        path_output_rec_weight=_resolve(base_dir, exp.get("path_output_rec_weight", "../results/{0}/weights/"), ds),
        path_output_rec_performance=_resolve(base_dir, exp.get("path_output_rec_performance",
                                                               "../results/{0}/performance/"), ds))
    if exp.get("negative_sampling"):
        # namespace_model.py:190-198: paths resolved against the config folder; strategy "random" writes its draws to
        # ../data/<dataset>/negative.tsv.  The candidate sets themselves are built in load_data (dataset/negative_sampling.py).
        nsd = {k: (_resolve(base_dir, v, ds) if isinstance(v, str) and k in ("files", "file_path") else
                   [_resolve(base_dir, x, ds) for x in v] if isinstance(v, list) else v) for k, v in exp["negative_sampling"].items()}
        if nsd.get("strategy") == "random":
            nsd["file_path"] = os.path.abspath(os.sep.join([base_dir, "..", "data", ds, "negative.tsv"]))
            os.makedirs(os.path.dirname(nsd["file_path"]), exist_ok=True)
        cfg.negative_sampling = _ns(nsd)
    for p in (cfg.path_output_rec_result, cfg.path_output_rec_weight, cfg.path_output_rec_performance):
        os.makedirs(p, exist_ok=True)
    return cfg


def _to_ns(x):
    if isinstance(x, dict):
        return SimpleNamespace(**{k: _to_ns(v) for k, v in x.items()})
    return x


def load_data_objects(exp, cfg, base_dir):
    """DataSetLoader(config).generate_dataobjects() (elliot/run.py:59-60): one list per test fold, one DataSet per validation fold;
    prefiltering, binarize and every splitting strategy of the reference (dataset/dataloader.py)."""
    from .dataset import dataloader, negative_sampling
    dc = dict(exp["data_config"])
    cfg.data_config = SimpleNamespace(**dc)
    cfg.random_seed = exp.get("random_seed", 42)
    cfg.binarize = bool(exp.get("binarize", False))
    if exp.get("prefiltering"):
        pf = exp["prefiltering"]
        cfg.prefiltering = [_to_ns(x) for x in (pf if isinstance(pf, list) else [pf])]      # namespace_model.py:177-182
    cfg.splitting = _to_ns(exp.get("splitting", {"test_splitting": {"strategy": "random_subsampling", "test_ratio": 0.2}}))
    objs = dataloader.DataSetLoader(cfg, resolve=lambda p: _resolve(base_dir, p, cfg.dataset)).generate_dataobjects()
    if hasattr(cfg, "negative_sampling"):                       # dataset.py:221-243
        import random
        rng = random.Random(42)         # negative_sampling.py:16 seeds `random` once per process: ONE stream over all data objects
        for fold in objs:
            for data in fold:
                negative_sampling.attach(data, cfg.negative_sampling, rng=rng)
    return objs


def load_data(exp, cfg, base_dir):
    """The first (test fold, validation fold) data set -- what a configuration without folds produces."""
    return load_data_objects(exp, cfg, base_dir)[0][0]


def model_params(model_cfg):
    meta = _ns(dict(model_cfg.get("meta", {})))
    params = {k: v for k, v in model_cfg.items() if k != "meta"}
    for k, v in params.items():
        if isinstance(v, (list, tuple)):
            raise Exception(f"hyper-parameter search ({k}: {v}) needs Elliot's driver; give scalars to the mini runner")
    ns = _ns(params)
    ns.meta = meta
    return ns


def run_experiment(config_path=""):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(message)s")
    with open(config_path) as f:
        exp = yaml.safe_load(f)["experiment"]
    base_dir = os.path.dirname(os.path.abspath(config_path))
    cfg = build_config(exp, base_dir)
    folds = load_data_objects(exp, cfg, base_dir)
    results = {}
    for key, model_cfg in exp["models"].items():
        cls = getattr(rec, key.split(".")[-1], None)
        if cls is None:
            raise Exception(f"Model {key} is not provided by elliot_amd (available: {rec.__all__})")
        for t_fold, val_folds in enumerate(folds):               # elliot/run.py:59-75: every model on every data object
            for v_fold, data in enumerate(val_folds):
                model = cls(data=data, config=cfg, params=model_params(model_cfg or {}))
                model.train()
                best = model.get_results()
                tag = model.name if (t_fold, v_fold) == (0, 0) else f"{model.name}#test{t_fold}val{v_fold}"
                results[tag] = best
                with open(os.path.join(cfg.path_output_rec_performance, f"rec_{tag}.tsv"), "w") as out:
                    for cutoff, d in best.items():
                        for metric, value in d["test_results"].items():
                            out.write(f"{model.name}\t{cutoff}\t{metric}\t{value}\n")
                print(f"{tag}: " + ", ".join(f"{m}@{c}={v:.5f}" for c, d in best.items() for m, v in d["test_results"].items()))
    return results


if __name__ == "__main__":
    run_experiment(sys.argv[1])
