"""elliot_amd -- MI355X (gfx950) backend for the latent-factor hot path of sisinflab/elliot.

Layout:
  csrc/          hand-written HIP kernels + the C ABI (include/elliot_hip.h) -> libelliot_hip.so
  _lib.py        ctypes binding of the C ABI (fails loudly when the .so is missing)
  ops.py         thin tensor-level wrappers (torch tensors are device-buffer holders only)
  recommender/   host-side mirror of Elliot's plugin surface (RecMixin / BaseRecommenderModel)
  external/      package loadable through Elliot's `external_models_path` mechanism
"""
__version__ = "0.1.0"
