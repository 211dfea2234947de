"""tensorflow.keras of the stand-in (see ../__init__.py): Model / Layer bookkeeping, the four layer types and the loss the
reference's model files use.  TEST INFRASTRUCTURE."""
import torch

import tensorflow as tf
from tensorflow import Tensor, Variable, _lookup, _raw, tf_clauses

from . import layers  # noqa: F401
from .layers import Layer


class Model(Layer):
    pass


class Sequential(Layer):
    def __init__(self, name=None):
        super().__init__(name=name or "sequential")
        self.layers = []

    def add(self, layer):
        self.layers.append(layer)

    def call(self, x, training=None):
        for layer in self.layers:
            x = layer(x, training=training) if isinstance(layer, layers.Dropout) else layer(x)
        return x


class _Backend:
    @staticmethod
    def random_normal(shape, mean=0.0, stddev=1.0):
        return Tensor(torch.randn(tuple(int(s) for s in shape)) * stddev + mean)

    @staticmethod
    def l2_normalize(x, axis=None):
        """[clause l2_normalize_epsilon_1e12_inside_max] x * rsqrt(max(sum(x^2), 1e-12))"""
        t = _raw(x)
        ss = (t * t).sum(dim=axis, keepdim=True)
        if tf_clauses.get("l2_normalize_epsilon_1e12_inside_max"):
            return Tensor(t * torch.rsqrt(torch.clamp(ss, min=1e-12)))
        return Tensor(t / (torch.sqrt(ss) + 1e-12))

    @staticmethod
    def epsilon():
        return 1e-7


backend = _Backend()


class _Activations:
    @staticmethod
    def sigmoid(x):
        return Tensor(torch.sigmoid(_raw(x)))

    @staticmethod
    def linear(x):
        return x


activations = _Activations()


class _BinaryCrossentropy:
    """keras.losses.BinaryCrossentropy() (from_logits=False): K.binary_crossentropy clips the probabilities to [1e-7, 1 - 1e-7]
    with clip_by_value (its gradient rule applies), -[y log(p + 1e-7) + (1 - y) log(1 - p + 1e-7)], mean over the last axis, then
    over the batch; a y_pred of rank one above y_true loses its last axis first (losses_utils.squeeze_or_expand_dimensions)."""

    def __call__(self, y_true, y_pred):
        p, y = _raw(y_pred), _raw(y_true).to(torch.float32)
        if p.dim() == y.dim() + 1 and p.shape[-1] == 1:
            p = p.squeeze(-1)
        if tf_clauses.get("bce_clips_probabilities_at_1e7"):
            p = tf.clip_by_value(Tensor(p), 1e-7, 1.0 - 1e-7).t
        eps = 1e-7 if tf_clauses.get("bce_adds_epsilon_inside_log") else 0.0
        bce = y * torch.log(p + eps) + (1 - y) * torch.log(1 - p + eps)
        return Tensor((-bce).mean())


class _MeanSquaredError:
    def __call__(self, y_true, y_pred):
        p, y = _raw(y_pred), _raw(y_true).to(torch.float32)
        if p.dim() == y.dim() + 1 and p.shape[-1] == 1:
            p = p.squeeze(-1)
        return Tensor(((p - y) ** 2).mean())


class _Losses:
    BinaryCrossentropy = _BinaryCrossentropy
    MeanSquaredError = _MeanSquaredError


losses = _Losses()


class _Regularizers:
    @staticmethod
    def l2(l=0.01):  # noqa: E741
        return ("l2", l)          # Dense(kernel_regularizer=...) only fills layer.losses, which the model files never read


regularizers = _Regularizers()
initializers = tf.initializers
