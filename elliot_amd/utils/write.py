"""Recommendation list writer -- same on-disk format as elliot/utils/write.py:35-44
(`user<TAB>item<TAB>score` per line, users in dict order, items in rank order)."""
import numpy as np


def store_recommendation(recommendations, path=""):
    with open(path, "w") as out:
        for u, recs in recommendations.items():
            out.writelines(f"{u}\t{i}\t{v}\n" for i, v in recs)


def store_recommendation_arrays(public_users, public_items, idx, val, path):
    """Array fast path: idx/val are [n_users, k] host arrays of private item ids / scores."""
    items = np.asarray(public_items)[idx]
    with open(path, "w") as out:
        for r, u in enumerate(public_users):
            out.writelines(f"{u}\t{items[r, c]}\t{val[r, c]}\n" for c in range(idx.shape[1]))
