"""Generate tests/golden/proxy_ref.json by RUNNING THE REFERENCE'S ProxyRecommender methods (build container only).

TEST INFRASTRUCTURE.  elliot/recommender/generic/Proxy/Proxy.py is loaded by file path with its three package imports
stubbed (importing `elliot.recommender` pulls TensorFlow); `read_recommendations` (:68-75) and
`get_single_recommendation` (:50-66) are then called unbound on a namespace carrying what they read (`_data`,
`_recommendations`).  The input file exercises: shuffled lines, equal scores, items of the training set, ids the dataset
does not know, users without lines, more lines than k.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_proxy.py
"""
import importlib.util
import json
import os
import sys
import types
from types import SimpleNamespace

sys.dont_write_bytecode = True
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def load_proxy():
    for name in ("elliot", "elliot.recommender", "elliot.recommender.base_recommender_model",
                 "elliot.recommender.recommender_utils_mixin"):
        sys.modules.setdefault(name, types.ModuleType(name))
    base = sys.modules["elliot.recommender.base_recommender_model"]
    base.BaseRecommenderModel = type("BaseRecommenderModel", (), {})
    base.init_charger = lambda f: f
    sys.modules["elliot.recommender.recommender_utils_mixin"].RecMixin = type("RecMixin", (), {})
    spec = importlib.util.spec_from_file_location("ref_proxy", os.path.join(REF, "elliot/recommender/generic/Proxy/Proxy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.ProxyRecommender


def main():
    rs = np.random.RandomState(12)
    U, I, k = 40, 60, 6
    train = rs.rand(U, I) < 0.15
    cand = rs.rand(U, I) < 0.4
    pub_u = {u: 100 + 3 * u for u in range(U)}            # private -> public
    pub_i = {i: 9000 + 7 * i for i in range(I)}
    lines = []
    for u in range(U):
        if u % 9 == 4:
            continue                                       # user without lines
        n = rs.randint(1, 15)
        items = rs.choice(I, n, replace=False)
        scores = np.round(rs.rand(n) * 4, 1)               # one decimal -> plenty of equal scores
        for i, s in zip(items, scores):
            lines.append((pub_u[u], pub_i[int(i)], float(s)))
    lines.append((pub_u[3], 123456, 3.9))                  # unknown item
    lines.append((77777, pub_i[5], 1.0))                   # unknown user
    order = rs.permutation(len(lines))
    lines = [lines[j] for j in order]
    tsv = os.path.join(OUT, "proxy_recs.tsv")
    with open(tsv, "w") as f:
        for u, i, s in lines:
            f.write(f"{u}\t{i}\t{s}\n")

    Proxy = load_proxy()
    data = SimpleNamespace(private_users=pub_u, private_items=pub_i)     # Elliot: private_users[private id] = public id
    me = SimpleNamespace(_data=data)
    me._recommendations = Proxy.read_recommendations(me, tsv)
    # pandas >= 2 hands groupby(['userId']) keys over as 1-tuples; the reference's pinned pandas gives scalars
    me._recommendations = {(u[0] if isinstance(u, tuple) else u): r for u, r in me._recommendations.items()}
    expected = {}
    for tag, mask in (("allunrated", ~train), ("candidates", cand)):
        me2 = SimpleNamespace(_data=data, _recommendations={u: r for u, r in me._recommendations.items()
                                                            if u in set(pub_u.values())})
        # the reference indexes candidate_items[u] for every user of the file: unknown users / items raise KeyError there,
        # so they are removed from ITS input; the mirror drops them itself.
        me2._recommendations = {u: [(i, p) for i, p in r if i in set(pub_i.values())] for u, r in me2._recommendations.items()}
        got = Proxy.get_single_recommendation(me2, mask, k)
        expected[tag] = {str(int(u)): [[int(i), float(p)] for i, p in r] for u, r in got.items()}
    json.dump({"U": U, "I": I, "k": k, "train": np.argwhere(train).tolist(), "cand": np.argwhere(cand).tolist(),
               "expected": expected}, open(os.path.join(OUT, "proxy_ref.json"), "w"))
    print("proxy_ref.json:", {t: sum(len(v) for v in e.values()) for t, e in expected.items()}, "kept rows")


if __name__ == "__main__":
    main()
