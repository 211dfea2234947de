"""Early stopping with the semantics of elliot/recommender/early_stopping.py:8-137.

Options (all optional, under the model's `early_stopping:` key): monitor (metric[@k] | "loss"), patience,
mode, min_delta, rel_delta, baseline, verbose.  With no options the criterion is inactive.  `stop()` looks at
the last patience+1 observations and fires when EVERY consecutive pair got worse (or stayed within
min_delta / rel_delta, or is on the wrong side of `baseline`).
"""


class EarlyStopping:
    def __init__(self, early_stopping_ns, validation_metric, validation_k, cutoffs, simple_metrics):
        opts = dict(vars(early_stopping_ns))
        self.validation_metric, self.validation_k = validation_metric, validation_k
        self.cutoffs, self.simple_metrics = cutoffs, simple_metrics
        self.monitor = opts.get("monitor", validation_metric)
        self.active = bool(opts)
        if not self.active:
            return
        self.patience = opts.get("patience", 0)
        if self.monitor == "loss":
            self.mode, self.metric = "min", False
        else:
            self.mode = "max"
            parts = self.monitor.split("@")
            if parts[0].lower() not in [m.lower() for m in simple_metrics]:
                raise Exception("Early stopping metric must be in the list of simple metrics")
            self.metric_k = int(parts[1]) if len(parts) > 1 else validation_k
            if self.metric_k not in cutoffs:
                raise Exception("Validation cutoff must be in general cutoff values")
            self.metric = parts[0]
        if "mode" in opts and opts["mode"] != "auto":
            self.mode = opts["mode"]
        for key in ("min_delta", "rel_delta", "baseline"):
            if key in opts:
                setattr(self, key, opts[key])
        self.verbose = opts.get("verbose", False)

    def _worse(self, older, newer):
        """True when `newer` did not improve on `older` in the sense of the configured deltas (values are
        presented so that 'older < newer' means degradation, as the reference orders them)."""
        if newer > older:
            return True
        if hasattr(self, "min_delta") and (older - newer) <= self.min_delta:
            return True
        if hasattr(self, "rel_delta") and (older - newer) <= older * self.rel_delta:
            return True
        if hasattr(self, "baseline"):
            if self.mode == "min":
                return older >= self.baseline
            if self.mode == "max":
                return older <= self.baseline
            raise ValueError("mode option must be in the list [min, max, auto]")
        return False

    def stop(self, losses, results):
        if not self.active:
            return False
        if not self.metric:
            observed = list(losses)
        else:
            observed = [r[self.metric_k]["val_results"][self.metric] for r in results]
        if len(observed) <= self.patience:
            return False
        window = observed[-(self.patience + 1):][::-1]      # newest first
        if self.mode == "min":
            window = window[::-1]                            # oldest first for a quantity we minimise
        checks = [self._worse(window[p], window[p + 1]) for p in range(len(window) - 1)]
        return bool(checks) and all(checks)

    def __str__(self):
        return ", ".join(f"{k}: {v}" for k, v in self.__dict__.items())
