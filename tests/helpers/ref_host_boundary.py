"""Run by tests/test_host_reference_boundary.py in a subprocess with PYTHONPATH=/root/reference (PYTHONDONTWRITEBYTECODE=1):
the plugin surface of elliot_amd inside the REAL host framework -- the reference's own DataSetLoader / Splitter / DataSet
(built from a synthetic TSV), its logging (`init`, `prepare_logger`), Evaluator, build_model_folder and
store_recommendation -- i.e. the HAVE_ELLIOT branch of elliot_amd/recommender/_compat.py, driven the way
elliot/run.py:61-75 and hyperoptimization/model_coordinator.py:62-65 drive a model class.  No GPU: the inner `_model` is a
stub that ranks by a fixed score table (the kernels have their own parity tests); what is exercised is every host-side
contract between our RecMixin / BaseRecommenderModel / init_charger and Elliot's services.  Prints one JSON line."""
import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    tmp = tempfile.mkdtemp(prefix="el_host_")
    rs = np.random.RandomState(0)
    rows = set()
    while len(rows) < 4000:
        rows.add((int(rs.randint(1000, 1120)), int(rs.randint(5000, 5300))))
    with open(os.path.join(tmp, "dataset.tsv"), "w") as f:
        for u, i in sorted(rows):
            f.write(f"{u}\t{i}\t{rs.randint(1, 6)}\t{rs.randint(0, 10**6)}\n")

    import elliot.utils.logging as elog                       # the reference's
    from elliot.dataset.dataset import DataSetLoader
    from elliot.evaluation.evaluator import Evaluator as RefEvaluator
    from elliot.utils.write import store_recommendation as ref_store

    for d in ("log", "recs", "weights"):
        os.makedirs(os.path.join(tmp, d), exist_ok=True)
    config = SimpleNamespace(
        config_test=False, binarize=False, random_seed=42, top_k=10, align_side_with_train=False,
        data_config=SimpleNamespace(strategy="dataset", dataset_path=os.path.join(tmp, "dataset.tsv"), side_information=[]),
        splitting=SimpleNamespace(test_splitting=SimpleNamespace(strategy="random_subsampling", test_ratio=0.2)),
        evaluation=SimpleNamespace(cutoffs=[10, 5], simple_metrics=["nDCG", "Precision", "Recall"], relevance_threshold=0,
                                   paired_ttest=False, wilcoxon_test=False, complex_metrics=[]),
        path_output_rec_result=os.path.join(tmp, "recs"), path_output_rec_weight=os.path.join(tmp, "weights"),
        path_log_folder=os.path.join(tmp, "log"))
    elog.init(os.path.join("/root/reference", "elliot", "config", "logger_config.yml"), config.path_log_folder)
    data = DataSetLoader(config=config).generate_dataobjects()[0][0]     # run.py:59-60

    from elliot_amd.recommender import _compat
    assert _compat.HAVE_ELLIOT and _compat.Evaluator is RefEvaluator and _compat.store_recommendation is ref_store
    from elliot_amd.recommender.base_recommender_model import BaseRecommenderModel, init_charger, param
    from elliot_amd.recommender.recommender_utils_mixin import RecMixin

    class StubScores:
        """Inner `_model`: a fixed [U, I] score table ranked under the tagged CSR mask get_candidate_mask() hands over
        (("excl", CSR) = every item NOT in the row, the allunrated_mask semantics of dataset.py:245)."""

        def __init__(self, U, I):
            self.ctx = SimpleNamespace(device=torch.device("cpu"))
            self.S = np.random.RandomState(7).normal(size=(U, I)).astype(np.float32)
            self.saved = None

        def recommend(self, mask, k, start, stop):
            kind, csr = mask
            S = self.S[start:stop].copy()
            ip, ix = csr.indptr.numpy(), csr.indices.numpy()
            allowed = np.zeros_like(S, dtype=bool)
            for r, u in enumerate(range(start, stop)):
                allowed[r, ix[ip[u]:ip[u + 1]]] = True
            if kind == "excl":
                allowed = ~allowed
            S[~allowed] = -np.inf
            idx = np.argsort(-S, axis=1, kind="stable")[:, :k].astype(np.int32)
            return torch.from_numpy(idx), torch.from_numpy(np.take_along_axis(S, idx, 1))

        def save_weights(self, path):
            self.saved = path

    class StubModel(RecMixin, BaseRecommenderModel):
        @init_charger
        def __init__(self, data, config, params, *args, **kwargs):
            self._params_list = [param("factors", "f", 8)]
            self.autoset_params()
            self._model = StubScores(self._num_users, self._num_items)
            self._sampler = None

        @property
        def name(self):
            return "StubModel_" + self.get_base_params_shortcut() + "_" + self.get_params_shortcut()

    key = "StubModel"
    elog.prepare_logger(key, config.path_log_folder)                      # run.py:66
    params = SimpleNamespace(meta=SimpleNamespace(save_recs=True, save_weights=True, validation_metric="nDCG@10"),
                             epochs=1, batch_size=64, seed=42, factors=8)
    model = StubModel(data=data, config=config, params=params)            # model_coordinator.py:62
    assert isinstance(model.evaluator, RefEvaluator)
    model.evaluate(it=0, loss=1.5)                                       # -> get_recommendations -> genuine Evaluator.eval -> recs TSV
    res = model.get_results()
    val_recs, test_recs = model.get_recommendations(10)
    # the dicts are what the reference's own get_single_recommendation would have produced for the same scores
    S, mask = model._model.S, data.allunrated_mask
    exp = {}
    for u in range(data.num_users):
        sc = np.where(mask[u], S[u], -np.inf)
        top = np.argsort(-sc, kind="stable")[:10]
        exp[data.private_users[u]] = [(data.private_items[int(i)], float(sc[i])) for i in top]
    same = all(test_recs[u] == exp[u] for u in exp) and set(test_recs) == set(exp)
    ref_eval = RefEvaluator(data, params).eval((exp, exp))
    # our array data plane (el_host_split_flags / el_host_pyset_order / vectorised CSR) on the same TSV vs the reference's
    # DataSetLoader + Splitter + DataSet: same split, same private ids, same matrices
    from elliot_amd.dataset.dataset import load_tsv_dataset, default_config
    ours = load_tsv_dataset(default_config(), config.data_config.dataset_path, 0.2, 42)
    ref_test = data.get_test()
    plane = {
        "users": list(ours.users) == list(data.users), "items": list(ours.items) == list(data.items),
        "transactions": int(ours.transactions) == int(data.transactions),
        "sp_i_train": bool((ours.sp_i_train != data.sp_i_train).nnz == 0),
        "test_dict": {u: dict(v) for u, v in ours.test_dict.items() if v} == {u: dict(v) for u, v in ref_test.items() if v},
        "train_dict": {u: dict(v) for u, v in ours.train_dict.items()} == {u: dict(v) for u, v in data.train_dict.items()},
    }
    ip, cols, vals = ours.split_csr(False)
    pu, pi = data.public_users, data.public_items
    csr_ok = True
    for u, its in ref_test.items():
        if u not in pu:
            continue
        row = {int(c): float(v) for c, v in zip(cols[ip[pu[u]]:ip[pu[u] + 1]], vals[ip[pu[u]]:ip[pu[u] + 1]]) if c < data.num_items}
        csr_ok &= row == {pi[i]: float(r) for i, r in its.items() if i in pi}
    plane["split_csr"] = bool(csr_ok)
    files = sorted(os.listdir(config.path_output_rec_result))
    first = open(os.path.join(config.path_output_rec_result, files[0])).readline().rstrip("\n").split("\t")
    out = {
        "have_elliot": bool(_compat.HAVE_ELLIOT), "users": int(data.num_users), "items": int(data.num_items),
        "recs_equal_reference_semantics": bool(same),
        "ndcg10": float(res[10]["test_results"]["nDCG"]), "ndcg10_reference_on_expected": float(ref_eval[10]["test_results"]["nDCG"]),
        "precision5": float(res[5]["test_results"]["Precision"]),
        "rec_files": files, "first_rec_row_fields": len(first), "first_rec_user": first[0],
        "weights_saved_to": os.path.basename(model._model.saved or ""), "weight_dir_exists": os.path.isdir(os.path.join(config.path_output_rec_weight, model.name)),
        "loss": float(model.get_loss()), "best_iteration": int(getattr(params, "best_iteration", -1)), "name": params.name,
        "data_plane": plane,
        "signature": list(__import__("inspect").signature(RecMixin.get_single_recommendation).parameters),
    }
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
