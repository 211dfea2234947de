"""tensorflow.keras.layers of the stand-in (see ../__init__.py).  TEST INFRASTRUCTURE."""
import torch

import tensorflow as tf
from tensorflow import Tensor, Variable, _lookup, _raw, tf_clauses

_UID = {}


def _auto_name(base):
    n = _UID.get(base, 0)
    _UID[base] = n + 1
    return base if n == 0 else f"{base}_{n}"


class Layer:
    """Attribute tracking in assignment order; trainable_weights = the layer's own variables, then its sub-layers' (what
    Layer.trainable_weights of TF 2.3 returns: self._trainable_weights + the children's, de-duplicated)."""

    def __init__(self, name=None, **kwargs):
        object.__setattr__(self, "_own", [])
        object.__setattr__(self, "_children", [])
        self.name = name or _auto_name(type(self).__name__.lower())

    def __setattr__(self, k, v):
        if isinstance(v, Variable) and all(v is not o for o in self._own):
            self._own.append(v)
        elif isinstance(v, Layer) and all(v is not o for o in self._children):
            self._children.append(v)
        object.__setattr__(self, k, v)

    @property
    def trainable_weights(self):
        out = list(self._own)
        for c in self._children + list(getattr(self, "layers", [])):
            for w in c.trainable_weights:
                if all(w is not o for o in out):
                    out.append(w)
        return out

    trainable_variables = trainable_weights

    def __call__(self, *args, **kwargs):
        return self.call(*args, **kwargs)


class Dense(Layer):
    def __init__(self, units, activation=None, kernel_initializer=None, kernel_regularizer=None, input_dim=None, name=None, **kw):
        super().__init__(name=name or _auto_name("dense"))
        self.units, self.activation = int(units), activation
        self.kernel_initializer = kernel_initializer or tf.initializers.GlorotUniform()
        self.kernel = self.bias = None
        if input_dim is not None:
            self._build(int(input_dim))

    def _build(self, n_in):
        self.kernel = Variable(self.kernel_initializer((n_in, self.units)), name=f"{self.name}/kernel")
        self.bias = Variable(torch.zeros(self.units), name=f"{self.name}/bias")

    def call(self, x, **kw):
        t = _raw(x)
        if self.kernel is None:
            self._build(t.shape[-1])
        y = t @ self.kernel.t + self.bias.t
        if self.activation in ("tanh",):
            y = torch.tanh(y)
        elif self.activation in ("relu",):
            y = torch.relu(y)
        elif self.activation in ("sigmoid",):
            y = torch.sigmoid(y)
        elif callable(self.activation):
            return self.activation(Tensor(y))
        elif self.activation is not None:
            raise NotImplementedError(self.activation)
        return Tensor(y)


class Dropout(Layer):
    """[clause dropout_scales_kept_units] training only: kept units scaled by 1 / (1 - rate)."""

    def __init__(self, rate, name=None, **kw):
        super().__init__(name=name or _auto_name("dropout"))
        self.rate = float(rate)

    def call(self, x, training=None, **kw):
        if not training or self.rate <= 0.0:
            return x if isinstance(x, Tensor) else Tensor(_raw(x))
        t = _raw(x)
        keep = (torch.rand(t.shape) >= self.rate).to(t.dtype)
        return Tensor(t * keep / (1.0 - self.rate) if tf_clauses.get("dropout_scales_kept_units") else t * keep)


class Lambda(Layer):
    def __init__(self, fn, name=None, **kw):
        super().__init__(name=name or _auto_name("lambda"))
        self.fn = fn

    def call(self, x, *args, **kw):                     # (the VAE passes a stray positional `1`: Keras reads it as `mask`)
        return self.fn(x)


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, embeddings_initializer=None, name=None, dtype=None, **kw):
        super().__init__(name=name or _auto_name("embedding"))
        self.input_dim, self.output_dim = int(input_dim), int(output_dim)
        self.embeddings_initializer = embeddings_initializer or tf.initializers.GlorotUniform()
        self.embeddings = None

    def call(self, ids, **kw):
        if self.embeddings is None:
            self.embeddings = Variable(self.embeddings_initializer((self.input_dim, self.output_dim)), name=f"{self.name}/embeddings")
        return _lookup(self.embeddings, ids)            # embedding_lookup: the gradient is an IndexedSlices
