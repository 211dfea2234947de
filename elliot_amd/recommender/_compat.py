"""Resolution of the Elliot services the plugin surface depends on.

Inside an Elliot installation the genuine Evaluator / folders / recommendation writer are used, so the
models behave exactly like in-tree Elliot models; stand-alone the API-compatible mirrors of this
package are used.  (This is host plumbing only: no numeric path depends on it.)
"""
try:  # inside an Elliot process (exercised by tests/test_host_reference_boundary.py with PYTHONPATH=/root/reference)
    from elliot.evaluation.evaluator import Evaluator
    from elliot.utils.folder import build_model_folder
    from elliot.utils.write import store_recommendation
    HAVE_ELLIOT = True
except Exception:
    from ..evaluation.evaluator import Evaluator
    from ..utils.folder import build_model_folder
    from ..utils.write import store_recommendation
    HAVE_ELLIOT = False

from ..utils import logging  # noqa: E402,F401
