"""Plugin surface, part 1: BaseRecommenderModel + @init_charger.

API-compatible mirror of elliot/recommender/base_recommender_model.py:27-163.  The experiment driver
(elliot/hyperoptimization/model_coordinator.py:62-65,100-103) needs exactly: the constructor
`cls(data=, config=, params=)`, `train()`, `get_loss()`, `get_results()`, `get_params()` and `name`;
everything else here exists so that model classes written for Elliot's recipe
(docs/source/guide/new_alg.rst:7-24: `_params_list` + `autoset_params()`) work unchanged.
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


class BaseRecommenderModel(ABC):
    def __init__(self, data, config, params, *args, **kwargs):
        self._data, self._config, self._params = data, config, params
        meta = params.meta
        ev = data.config.evaluation
        self._negative_sampling = hasattr(data.config, "negative_sampling")
        self._restore = getattr(meta, "restore", False)

        cutoffs = getattr(ev, "cutoffs", [data.config.top_k])
        cutoffs = cutoffs if isinstance(cutoffs, list) else [cutoffs]
        first_metric = ev.simple_metrics[0] if ev.simple_metrics else ""
        vm = getattr(meta, "validation_metric", f"{first_metric}@{cutoffs[0]}").split("@")
        if vm[0].lower() not in [m.lower() for m in ev.simple_metrics]:
            raise Exception("Validation metric must be in the list of simple metrics")
        self._validation_k = int(vm[1]) if len(vm) > 1 else cutoffs[0]
        if self._validation_k not in cutoffs:
            raise Exception("Validation cutoff must be in general cutoff values")
        self._validation_metric = vm[0]

        self._save_weights = getattr(meta, "save_weights", False)
        self._save_recs = getattr(meta, "save_recs", False)
        self._verbose = getattr(meta, "verbose", None)
        self._validation_rate = getattr(meta, "validation_rate", 1)
        self._optimize_internal_loss = getattr(meta, "optimize_internal_loss", False)
        self._epochs = int(getattr(params, "epochs", 2))
        self._seed = getattr(params, "seed", 42)
        self._early_stopping = EarlyStopping(SimpleNamespace(**getattr(params, "early_stopping", {})),
                                             self._validation_metric, self._validation_k, cutoffs, ev.simple_metrics)
        self._iteration = 0
        if self._epochs < self._validation_rate:
            raise Exception(f"The first validation epoch ({self._validation_rate}) "
                            f"is later than the overall number of epochs ({self._epochs}).")
        self._batch_size = getattr(params, "batch_size", -1)
        self.best_metric_value = 0
        self._losses, self._results, self._params_list = [], [], []

    # -- naming helpers (the result files are keyed by these strings) --------------------------------
    def get_base_params_shortcut(self):
        base = {"seed": self._seed, "e": self._epochs, "bs": self._batch_size}
        return "_".join(f"{k}={str(v).replace('.', '$')}" for k, v in base.items())

    def get_params_shortcut(self):
        parts = []
        for attr, _public, short, _default, _read, show in self._params_list:
            v = getattr(self, attr)
            parts.append(f"{short}={str(show(v) if show else v).replace('.', '$')}")
        return "_".join(parts)

    def autoset_params(self):
        """Bind `(attr, yaml_key, shortcut, default, reader, printer)` tuples to attributes."""
        self.logger.info("Loading parameters")
        for attr, public, _short, default, read, _show in self._params_list:
            raw = getattr(self._params, public, default)
            setattr(self, attr, raw if read is None else read(raw))
            self.logger.info(f"Parameter {public} set to {getattr(self, attr)}")
        if not self._params_list:
            self.logger.info("No parameters defined")

    @staticmethod
    def _batch_remove(original_str, char_list):
        for c in char_list:
            original_str = original_str.replace(c, "")
        return original_str

    @abstractmethod
    def train(self):
        pass

    @abstractmethod
    def get_recommendations(self, *args):
        pass

    @abstractmethod
    def get_loss(self):
        pass

    @abstractmethod
    def get_params(self):
        pass

    @abstractmethod
    def get_results(self):
        pass


def init_charger(init):
    """Decorator of every model constructor (base_recommender_model.py:142-163): base init, logger, RNG
    seeding (np.random + random, :149-150), user init, then Evaluator and the weights folder."""
    @wraps(init)
    def new_init(self, *args, **kwargs):
        BaseRecommenderModel.__init__(self, *args, **kwargs)
        package = inspect.getmodule(self).__package__ or ""
        rec_name = f"external.{self.__class__.__name__}" if "external" in package else self.__class__.__name__
        level = pylog.CRITICAL if getattr(self._config, "config_test", False) else pylog.DEBUG
        self.logger = _compat.logging.get_logger_model(rec_name, level)
        np.random.seed(self._seed)
        random.seed(self._seed)
        self._nprandom, self._random = np.random, random
        self._num_items, self._num_users = self._data.num_items, self._data.num_users

        init(self, *args, **kwargs)

        self.evaluator = _compat.Evaluator(self._data, self._params)
        self._params.name = self.name
        _compat.build_model_folder(self._config.path_output_rec_weight, self.name)
        self._saving_filepath = os.path.abspath(
            os.sep.join([self._config.path_output_rec_weight, self.name, f"best-weights-{self.name}"]))
    return new_init
