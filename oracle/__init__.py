"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatements of the reference's latent-factor hot path (sisinflab/elliot v0.3.1), each function
citing the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this package; nothing under elliot_amd/ does.

Pinning status:
  * sampler.RefSampler, sgd.update_factors, sgd.get_user_predictions -- pinned against the
    reference's OWN code run in the build container (oracle/gen_golden.py -> tests/golden/*.npz).
  * bprmf_batch.* (TF model) -- "parity unpinned": TensorFlow 2.3.2 is not installable here and
    the reference ships no tests/golden vectors; pinned instead by hand-computed known-answer
    cases and an independent autograd derivation (tests/test_oracle_*.py).
"""
