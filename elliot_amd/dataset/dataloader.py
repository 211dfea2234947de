"""Ratings file -> prefilter -> split -> DataSet objects, on column arrays.

Mirrors the `dataset` / `fixed` strategies of elliot/dataset/dataset.py:28-135 (DataSetLoader), the prefilters of
elliot/prefiltering/standard_prefilters.py:15-199 and the splitting strategies of elliot/splitter/base_splitter.py:63-335:
same YAML options, same rows in train / validation / test -- computed with counting sorts and boolean masks instead of pandas
groupby + per-row `apply` (the reference's `splitting_temporal_holdout` evaluates a Python lambda per ROW).  Row order inside
every output is the file order, as in the reference (its filters and flag selections keep the frame order).

A "frame" here is a dict of equally long NumPy columns: userId, itemId, rating and (when the file has it) timestamp.
"""
import math
from types import SimpleNamespace

import numpy as np

from .dataset import DataSet, _split_flags, np_seed_state

COLUMNS = ("userId", "itemId", "rating", "timestamp")


def _get(ns, key, default=None):
    return ns.get(key, default) if isinstance(ns, dict) else getattr(ns, key, default)


def _has(ns, key):
    return (key in ns) if isinstance(ns, dict) else hasattr(ns, key)


def read_ratings(path):
    """dataset.py:99-104: tab-separated, no header, up to four columns; an all-missing timestamp column is dropped (check_timestamp)."""
    import pandas as pd
    df = pd.read_csv(path, sep="\t", header=None, names=list(COLUMNS))
    frame = {c: df[c].values for c in COLUMNS[:3]}
    if not df["timestamp"].isna().all():
        frame["timestamp"] = df["timestamp"].values
    return frame


def take(frame, keep):
    return {c: v[keep] for c, v in frame.items()}


def _dense_rank(values):
    """ids 0..n-1 in ascending value order (the pandas groupby order) + the group sizes."""
    values = np.asarray(values)
    if values.shape[0] and values.dtype.kind in "iu":
        lo = int(values.min())
        if int(values.max()) - lo < 8 * values.shape[0] + (1 << 24):
            cnt = np.bincount(values - lo)
            return (np.cumsum(cnt > 0) - 1)[values - lo], cnt[cnt > 0]
    uniq, inv, cnt = np.unique(values, return_inverse=True, return_counts=True)
    return inv, cnt


# ---- prefiltering (standard_prefilters.py) ---------------------------------------------------------------------------------
def _single_filter(frame, ns):
    strategy = _get(ns, "strategy")
    n = frame["userId"].shape[0]
    if strategy == "global_threshold":
        threshold = _get(ns, "threshold")
        if threshold is None:
            raise Exception("Threshold option is missing")
        if str(threshold).isdigit():
            return take(frame, frame["rating"] >= threshold)
        if threshold == "average":
            return take(frame, frame["rating"] >= frame["rating"].mean())
        raise Exception("Threshold value not recognized")
    if strategy == "user_average":
        r, cnt = _dense_rank(frame["userId"])
        mean = np.bincount(r, weights=frame["rating"].astype(np.float64), minlength=cnt.shape[0]) / cnt
        return take(frame, frame["rating"] >= mean[r])

    def core_value(name="core"):
        v = _get(ns, name)
        if v is None:
            raise Exception(f"{name.capitalize()} option is missing")
        if not str(v).isdigit():
            raise Exception(f"{name.capitalize()} option is not a digit")
        return int(v)

    def by_size(fr, column, pred):
        r, cnt = _dense_rank(fr[column])
        return take(fr, pred(cnt)[r])

    if strategy == "user_k_core":
        c = core_value()
        return by_size(frame, "userId", lambda cnt: cnt >= c)
    if strategy == "item_k_core":
        c = core_value()
        return by_size(frame, "itemId", lambda cnt: cnt >= c)
    if strategy == "iterative_k_core":
        c = core_value()
        while True:
            before = frame["userId"].shape[0]
            frame = by_size(by_size(frame, "userId", lambda cnt: cnt >= c), "itemId", lambda cnt: cnt >= c)
            if frame["userId"].shape[0] == before:
                return frame
    if strategy == "n_rounds_k_core":
        if _get(ns, "core") is None or _get(ns, "rounds") is None:
            raise Exception("Core or rounds options are missing")
        if not (str(_get(ns, "core")).isdigit() and str(_get(ns, "rounds")).isdigit()):
            raise Exception("Core or rounds options are not digits")
        c = int(_get(ns, "core"))
        for _ in range(int(_get(ns, "rounds"))):
            frame = by_size(by_size(frame, "userId", lambda cnt: cnt >= c), "itemId", lambda cnt: cnt >= c)
        return frame
    if strategy == "cold_users":
        t = _get(ns, "threshold")
        if t is None:
            raise Exception("Threshold option is missing")
        if not str(t).isdigit():
            raise Exception("Threshold option is not a digit")
        return by_size(frame, "userId", lambda cnt: cnt <= int(t))
    raise Exception("Misssing strategy")
    del n


def prefilter(frame, strategies):
    """PreFilter.filter: the configured strategies one after the other."""
    for ns in strategies or []:
        frame = _single_filter(frame, ns)
    return frame


# ---- splitting (base_splitter.py) -------------------------------------------------------------------------------------------
def _rank_first(frame, ascending):
    """data.groupby('userId')['timestamp'].rank(method='first', ascending=...) -- 1-based, ties in file order."""
    r, cnt = _dense_rank(frame["userId"])
    ts = np.asarray(frame["timestamp"])
    n = ts.shape[0]
    key = ts if ascending else -ts.astype(np.float64) if ts.dtype.kind == "f" else -ts.astype(np.int64)
    order = np.lexsort((np.arange(n), key, r))
    start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
    rank = np.empty(n, dtype=np.int64)
    rank[order] = np.arange(n) - np.repeat(start, cnt) + 1
    return rank, r, cnt


def _test_flags(frame, ns, seed, state=None):
    """[folds, rows] int8 test flags of ONE level of the hierarchy (handle_hierarchy, :134-196).  state: the np.random stream of
    the whole process_splitting call (the random strategies draw from it and leave it advanced)."""
    strategy = _get(ns, "strategy")
    if strategy is None:
        raise Exception("Strategy option not found")
    users = frame["userId"]
    if strategy == "fixed_timestamp":
        if not _has(ns, "timestamp"):
            raise Exception(f"Option timestamp missing for {strategy} strategy")
        t = str(_get(ns, "timestamp"))
        if t.isdigit():
            return (frame["timestamp"] >= int(t)).astype(np.int8)[None]
        if t == "best":
            return (frame["timestamp"] >= _best_timestamp(frame, int(_get(ns, "min_below", 1)), int(_get(ns, "min_over", 1)))).astype(np.int8)[None]
        raise Exception("Timestamp option value is not valid")
    if strategy == "temporal_hold_out":
        if _has(ns, "test_ratio"):
            rank, r, cnt = _rank_first(frame, True)
            ratio = float(_get(ns, "test_ratio"))
            thr = np.array([math.floor(x * (1 - ratio)) for x in cnt.tolist()], dtype=np.int64)     # :239, Python float arithmetic
            return (rank > thr[r]).astype(np.int8)[None]
        if _has(ns, "leave_n_out"):
            rank, _, _ = _rank_first(frame, False)
            return (rank <= int(_get(ns, "leave_n_out"))).astype(np.int8)[None]
        raise Exception(f"Option missing for {strategy} strategy")
    if strategy == "random_subsampling":
        folds = _get(ns, "folds", 1)
        if not str(folds).isdigit():
            raise Exception("Folds option value is not valid")
        if _has(ns, "test_ratio"):
            return _split_flags(users, 0, float(_get(ns, "test_ratio")), seed, int(folds), state=state)
        if _has(ns, "leave_n_out"):
            return _split_flags(users, 1, int(_get(ns, "leave_n_out")), seed, int(folds), state=state)
        raise Exception(f"Option missing for {strategy} strategy")
    if strategy == "random_cross_validation":
        if not _has(ns, "folds"):
            raise Exception(f"Option missing for {strategy} strategy")
        folds = _get(ns, "folds")
        if not str(folds).isdigit():
            raise Exception("Folds option value is not valid")
        folds = int(folds)
        r, cnt = _dense_rank(users)
        order = DataSet._group_stable(r, cnt.shape[0])
        start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
        pos = np.empty(users.shape[0], dtype=np.int64)
        pos[order] = np.arange(users.shape[0]) - np.repeat(start, cnt)            # position inside the user's rows, file order
        fold = pos % folds                                                         # fold_list_generator (:205-212)
        return (fold[None, :] == np.arange(folds)[:, None]).astype(np.int8)
    raise Exception(f"Unrecognized Test Strategy:\t{strategy}")


def _best_timestamp(frame, min_below, min_over):
    """splitting_best_timestamp (:296-318): the timestamp that leaves the most users with >= min_below rows before it and >=
    min_over rows from it on; ties -> the largest.  Per user the admissible timestamps form an interval of its sorted rows, so
    the count per candidate is a difference array over the distinct timestamps instead of a users x timestamps double loop."""
    ts = np.asarray(frame["timestamp"])
    uniq = np.unique(ts)
    r, cnt = _dense_rank(frame["userId"])
    order = np.lexsort((ts, r))
    st = ts[order]
    start = np.concatenate([[0], np.cumsum(cnt)])
    diff = np.zeros(uniq.shape[0] + 1, dtype=np.int64)
    ok = cnt >= min_below + min_over
    # below(t) = #rows < t >= min_below  <=>  t > st[min_below - 1];   over(t) = n - below(t) >= min_over  <=>  t <= st[n - min_over]
    for g in np.flatnonzero(ok).tolist():            # users are few compared with rows; each step is O(log T)
        a, b = start[g], start[g + 1]
        lo_t = st[a + min_below - 1] if min_below > 0 else None
        hi_t = st[b - min_over] if min_over > 0 else None
        lo = np.searchsorted(uniq, lo_t, side="right") if lo_t is not None else 0
        hi = np.searchsorted(uniq, hi_t, side="right") if hi_t is not None else uniq.shape[0]
        if hi > lo:
            diff[lo] += 1
            diff[hi] -= 1
    score = np.cumsum(diff[:-1])
    return uniq[np.flatnonzero(score == score.max())].max()


def split(frame, splitting, seed=42):
    """Splitter.process_splitting (:72-108): [(train, test), ...] or, with validation_splitting,
    [([(train, val), ...], test), ...] -- frames in file order."""
    if not _has(splitting, "test_splitting"):
        raise Exception("Test splitting strategy is not defined")

    state = np_seed_state(seed)              # np.random.seed(seed) ONCE (:73): every level and fold below continues this stream

    def level(fr, ns):
        flags = _test_flags(fr, ns, seed, state)
        return [(take(fr, f == 0), take(fr, f == 1)) for f in flags]

    out = level(frame, _get(splitting, "test_splitting"))
    if _has(splitting, "validation_splitting"):
        out = [(level(train, _get(splitting, "validation_splitting")), test) for train, test in out]
    return out


# ---- loader --------------------------------------------------------------------------------------------------------------------
def _triples(frame):
    return frame["userId"], frame["itemId"], frame["rating"]


class DataSetLoader:
    """generate_dataobjects() like dataset.py:137-150: one list per test fold, one DataSet per validation fold."""

    def __init__(self, config, resolve=lambda p: p):
        self.config = config
        dc = config.data_config
        strategy = _get(dc, "strategy")
        binarize = bool(getattr(config, "binarize", False))

        def prep(fr):
            if binarize or np.all(np.isnan(np.asarray(fr["rating"], dtype=np.float64))):
                fr = dict(fr, rating=np.ones(fr["userId"].shape[0], dtype=np.int64))
            return fr

        if strategy == "fixed":
            train, test = prep(read_ratings(resolve(_get(dc, "train_path")))), prep(read_ratings(resolve(_get(dc, "test_path"))))
            if _get(dc, "validation_path"):
                self.tuple_list = [([(train, prep(read_ratings(resolve(_get(dc, "validation_path")))))], test)]
            else:
                self.tuple_list = [(train, test)]
        elif strategy == "dataset":
            frame = read_ratings(resolve(_get(dc, "dataset_path")))
            frame = prefilter(frame, getattr(config, "prefiltering", None))
            self.tuple_list = split(prep(frame), config.splitting, getattr(config, "random_seed", 42))
        else:
            raise Exception("Strategy option not recognized")

    def generate_dataobjects(self):
        out = []
        for train_val, test in self.tuple_list:
            if isinstance(train_val, list):
                out.append([DataSet(self.config, _triples(train), _triples(test), _triples(val)) for train, val in train_val])
            else:
                out.append([DataSet(self.config, _triples(train_val), _triples(test))])
        return out
