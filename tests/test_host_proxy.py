"""ProxyRecommender mirror against the reference's own read_recommendations / get_single_recommendation
(fixture tests/golden/proxy_ref.json + proxy_recs.tsv, made by oracle/gen_golden_proxy.py), and a write -> read -> evaluate
round trip with the stand-alone evaluator.  No GPU involved."""
import json
import os
from types import SimpleNamespace

import numpy as np

from elliot_amd.dataset.dataset import DataSet, default_config
from elliot_amd.recommender import ProxyRecommender
from elliot_amd.utils.write import store_recommendation

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dataset(tmp_path, ref, with_candidates):
    U, I = ref["U"], ref["I"]
    tr = np.asarray(ref["train"])
    # every user and every item has to own a train row for the id maps; the fixture's ids: user 100+3u, item 9000+7i
    pad_u = np.setdiff1d(np.arange(U), tr[:, 0])
    pad_i = np.setdiff1d(np.arange(I), tr[:, 1])
    assert len(pad_u) == 0 and len(pad_i) == 0, "fixture must touch every user and item"
    pub_users, pub_items = 100 + 3 * np.arange(U), 9000 + 7 * np.arange(I)
    cfg = default_config(top_k=ref["k"], cutoffs=[ref["k"]], simple_metrics=["nDCG", "Recall"], out_dir=str(tmp_path))
    if with_candidates:
        cfg.negative_sampling = SimpleNamespace(strategy="fixed")
    for p in (cfg.path_output_rec_result, cfg.path_output_rec_weight):
        os.makedirs(p, exist_ok=True)
    te_u, te_i = (tr[:, 0] + 1) % U, (tr[:, 1] * 7 + 3) % I
    data = DataSet(cfg, (pub_users[tr[:, 0]], pub_items[tr[:, 1]], np.ones(len(tr))),
                   (pub_users[te_u], pub_items[te_i], np.ones(len(tr))), public_users=pub_users, public_items=pub_items)
    if with_candidates:
        c = np.zeros((U, I), bool)
        cc = np.asarray(ref["cand"])
        c[cc[:, 0], cc[:, 1]] = True
        data.test_mask = c
    return data, cfg


def _proxy(data, cfg, path):
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=1, path=path)
    return ProxyRecommender(data=data, config=cfg, params=params)


def test_reader_and_filter_equal_the_reference(tmp_path):
    ref = json.load(open(os.path.join(GOLD, "proxy_ref.json")))
    for tag, with_cand in (("allunrated", False), ("candidates", True)):
        data, cfg = _dataset(tmp_path, ref, with_cand)
        p = _proxy(data, cfg, os.path.join(GOLD, "proxy_recs.tsv"))
        assert p.name == "proxy_recs"
        p._table = p.read_recommendations(p._path)
        got = p.get_single_recommendation(p.get_candidate_mask(), ref["k"])
        exp = {int(u): [(i, s) for i, s in r] for u, r in ref["expected"][tag].items()}
        got = {u: r for u, r in got.items() if u in exp}              # the unknown user only exists on our side
        assert got == exp, tag


def test_write_read_evaluate_round_trip(tmp_path):
    ref = json.load(open(os.path.join(GOLD, "proxy_ref.json")))
    data, cfg = _dataset(tmp_path, ref, False)
    rs = np.random.RandomState(0)
    train = data.sp_i_train
    recs = {}
    for u in range(data.num_users):
        seen = set(train.indices[train.indptr[u]:train.indptr[u + 1]].tolist())
        free = [i for i in rs.permutation(data.num_items).tolist() if i not in seen][:ref["k"]]
        scores = np.sort(rs.rand(len(free)))[::-1]
        recs[data.private_users[u]] = [(data.private_items[i], float(s)) for i, s in zip(free, scores)]
    path = os.path.join(str(tmp_path), "model_x.tsv")
    store_recommendation(recs, path)
    p = _proxy(data, cfg, path)
    p.train()
    direct = p.evaluator.eval((recs, recs))
    got = p.get_results()
    assert p.name == "model_x"
    for k in direct:
        for m, v in direct[k]["test_results"].items():
            assert abs(got[k]["test_results"][m] - v) < 1e-12, (k, m)
    back = p.get_recommendations(ref["k"])[1]
    assert {u: [i for i, _ in r] for u, r in back.items()} == {u: [i for i, _ in r] for u, r in recs.items()}
