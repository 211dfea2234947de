"""Plugin surface, part 1: BaseRecommenderModel + @init_charger.

What Elliot's experiment driver needs from a model class (hyperoptimization/model_coordinator.py:62-65,100-103) is the
constructor `cls(data=, config=, params=)`, `train()`, `get_loss()`, `get_results()`, `get_params()` and `name`; what
model classes written to Elliot's recipe (docs/source/guide/new_alg.rst:7-24) need from their base is `_params_list` +
`autoset_params()`, the attributes listed in `_META_FIELDS` below, the two name helpers and the `init_charger`
decorator.  This module provides that contract (reference: recommender/base_recommender_model.py:27-163) in its own
arrangement: option tables instead of one assignment per option, one helper per concern.
"""
import inspect
import logging as pylog
import os
import random
from abc import ABC, abstractmethod
from functools import wraps
from types import SimpleNamespace

import numpy as np

from . import _compat
from .early_stopping import EarlyStopping

# attribute <- params.meta.<key> (default): the switches RecMixin and the model classes read
_META_FIELDS = (
    ("_restore", "restore", False),
    ("_save_weights", "save_weights", False),
    ("_save_recs", "save_recs", False),
    ("_verbose", "verbose", None),
    ("_validation_rate", "validation_rate", 1),
    ("_optimize_internal_loss", "optimize_internal_loss", False),
)


def param(key, short, default, read=None, show=None, attr=None):
    """One entry of a model's `_params_list`: (attribute, YAML key, file-name shortcut, default, reader, printer)."""
    return (attr or "_" + key, key, short, default, read, show)


def _as_list(x):
    return x if isinstance(x, list) else [x]


def _dollar(value):
    return str(value).replace(".", "$")


class BaseRecommenderModel(ABC):
    def __init__(self, data, config, params, *args, **kwargs):
        self._data, self._config, self._params = data, config, params
        evaluation = data.config.evaluation
        self._negative_sampling = hasattr(data.config, "negative_sampling")
        for attr, key, default in _META_FIELDS:
            setattr(self, attr, getattr(params.meta, key, default))
        cutoffs = _as_list(getattr(evaluation, "cutoffs", [data.config.top_k]))
        self._validation_metric, self._validation_k = self._validation_target(params.meta, evaluation, cutoffs)
        self._epochs = int(getattr(params, "epochs", 2))
        self._seed = getattr(params, "seed", 42)
        self._batch_size = getattr(params, "batch_size", -1)
        if self._epochs < self._validation_rate:
            raise Exception(f"The first validation epoch ({self._validation_rate}) "
                            f"is later than the overall number of epochs ({self._epochs}).")
        self._early_stopping = EarlyStopping(SimpleNamespace(**getattr(params, "early_stopping", {})),
                                             self._validation_metric, self._validation_k, cutoffs,
                                             evaluation.simple_metrics)
        self._iteration = 0
        self.best_metric_value = 0
        self._losses, self._results, self._params_list = [], [], []

    @staticmethod
    def _validation_target(meta, evaluation, cutoffs):
        """`validation_metric: nDCG@10` -> ("nDCG", 10); default = first simple metric at the first cutoff."""
        metrics = list(evaluation.simple_metrics)
        spec = getattr(meta, "validation_metric", f"{metrics[0] if metrics else ''}@{cutoffs[0]}")
        name, _, at = spec.partition("@")
        if name.lower() not in {m.lower() for m in metrics}:
            raise Exception("Validation metric must be in the list of simple metrics")
        k = int(at) if at else cutoffs[0]
        if k not in cutoffs:
            raise Exception("Validation cutoff must be in general cutoff values")
        return name, k

    # -- names of result / weight files ---------------------------------------------------------------------------
    def get_base_params_shortcut(self):
        return "_".join(f"{tag}={_dollar(v)}" for tag, v in (("seed", self._seed), ("e", self._epochs), ("bs", self._batch_size)))

    def get_params_shortcut(self):
        return "_".join(f"{short}={_dollar(show(getattr(self, attr)) if show else getattr(self, attr))}"
                        for attr, _key, short, _default, _read, show in self._params_list)

    def autoset_params(self):
        """Turn the `_params_list` entries into attributes (YAML value or default, passed through the reader)."""
        self.logger.info("Loading parameters")
        if not self._params_list:
            self.logger.info("No parameters defined")
        for attr, key, _short, default, read, _show in self._params_list:
            value = getattr(self._params, key, default)
            setattr(self, attr, read(value) if read else value)
            self.logger.info(f"Parameter {key} set to {getattr(self, attr)}")

    @staticmethod
    def _batch_remove(original_str, char_list):
        for piece in char_list:
            original_str = original_str.replace(piece, "")
        return original_str

    @abstractmethod
    def train(self):
        ...

    @abstractmethod
    def get_recommendations(self, *args):
        ...

    @abstractmethod
    def get_loss(self):
        ...

    @abstractmethod
    def get_params(self):
        ...

    @abstractmethod
    def get_results(self):
        ...


def init_charger(init):
    """Decorator every model constructor wears (reference :142-163).  Order matters and is part of the contract: base
    fields -> logger named after the YAML key -> np.random / random seeded with `seed` -> the model's own __init__ ->
    evaluator, `params.name`, weight folder and `_saving_filepath`."""
    @wraps(init)
    def charged(self, *args, **kwargs):
        BaseRecommenderModel.__init__(self, *args, **kwargs)
        key = self.__class__.__name__
        if "external" in (inspect.getmodule(self).__package__ or ""):
            key = "external." + key
        quiet = getattr(self._config, "config_test", False)
        self.logger = _compat.logging.get_logger_model(key, pylog.CRITICAL if quiet else pylog.DEBUG)
        for rng in (np.random, random):
            rng.seed(self._seed)
        self._nprandom, self._random = np.random, random
        self._num_users, self._num_items = self._data.num_users, self._data.num_items

        init(self, *args, **kwargs)

        self.evaluator = _compat.Evaluator(self._data, self._params)
        self._params.name = self.name
        weights_root = self._config.path_output_rec_weight
        _compat.build_model_folder(weights_root, self.name)
        self._saving_filepath = os.path.abspath(os.path.join(weights_root, self.name, f"best-weights-{self.name}"))
    return charged
