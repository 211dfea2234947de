#!/usr/bin/env python
"""Run the reference's UNMODIFIED TensorFlow model files on the `tensorflow` stand-in (oracle/tf_shim: torch-CPU autograd, the
library clauses of oracle/tf_clauses.py) and write tests/golden/tfshim_*.npz.  TEST INFRASTRUCTURE.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_tfshim.py [--reference /root/reference] [--out tests/golden]

The generators are the ones of oracle/gen_golden_tf.py (same injected weights, same clause-hitting batches, same file layout):
what differs is what `import tensorflow` resolves to inside the reference's model files.  The fixtures therefore pin the oracle's
restatement of the FILES -- BPRMF_batch_model.py:47-88, multi_vae_model.py:20-159, neural_matrix_factorization_model.py:75-148,
generalized_matrix_factorization_model.py:59-93 executed line by line from the reference's own source -- and nothing about
TensorFlow's library behaviour: that is oracle/gen_golden_tf.py's job under the real tensorflow==2.3.2 ("parity unpinned" until
someone runs it).  tests/test_tf_shim_pins.py consumes the files with the same checks (tests/helpers/tf_pins.py).
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="bprmf_batch,multivae,neumf,gmf,lightgcn,ngcf")
    args = ap.parse_args()
    sys.dont_write_bytecode = True                              # nothing may be written under the reference checkout
    sys.path.insert(0, os.path.join(HERE, "tf_shim"))           # `import tensorflow` -> the stand-in
    sys.path.insert(1, REPO)
    import tensorflow as tf
    assert "shim" in tf.__version__, "a real TensorFlow is importable here: run oracle/gen_golden_tf.py instead"
    import numpy as np
    from oracle import gen_golden_tf as g
    os.makedirs(args.out, exist_ok=True)
    todo = set(args.only.split(","))
    for name, fn in (("bprmf_batch", g.gen_bprmf_batch), ("multivae", g.gen_multivae), ("neumf", g.gen_neumf), ("gmf", g.gen_gmf),
                     ("lightgcn", g.gen_lightgcn), ("ngcf", g.gen_ngcf)):
        if name in todo:
            fn(args.reference, args.out, tf, prefix="tfshim_")
    with open(os.path.join(args.out, "tfshim_VERSION.txt"), "w") as f:
        f.write(f"tensorflow stand-in: {tf.__version__}\nnumpy {np.__version__}\n"
                f"reference model files executed unmodified from the sisinflab/elliot v0.3.1 checkout\n")


if __name__ == "__main__":
    main()
