"""The plugin surface inside the REAL host framework: the reference's own DataSet (loader + splitter on a
synthetic TSV), logging, Evaluator, folders and recommendation writer consume what our RecMixin / BaseRecommenderModel /
init_charger produce -- the HAVE_ELLIOT branch of elliot_amd/recommender/_compat.py, run in a subprocess with
PYTHONPATH=/root/reference.  Needs the reference checkout (build container); skipped where it does not exist (GPU box)."""
import inspect
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "elliot")), reason="reference checkout not present")
def test_plugin_surface_inside_the_reference_host():
    env = dict(os.environ, PYTHONPATH=REF + os.pathsep + REPO, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "tests", "helpers", "ref_host_boundary.py")], capture_output=True,
                         text=True, timeout=600, env=env, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:] + out.stdout[-1000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    r = json.loads(line[len("RESULT "):])
    assert r["have_elliot"] and r["users"] == 120 and r["items"] > 250
    assert r["recs_equal_reference_semantics"]                       # same dicts as recommender_utils_mixin.py:84-88 would build
    assert abs(r["ndcg10"] - r["ndcg10_reference_on_expected"]) < 1e-15 and 0 <= r["precision5"] <= 1
    assert len(r["rec_files"]) == 1 and r["rec_files"][0].endswith("_it=1.tsv") and r["first_rec_row_fields"] == 3
    assert r["weight_dir_exists"] and r["weights_saved_to"].startswith("best-weights-StubModel_")
    assert r["best_iteration"] == 1 and r["name"].startswith("StubModel_seed=42_e=1_bs=64")
    assert r["signature"] == ["self", "mask", "k", "predictions", "offset", "offset_stop"]
    # our TSV -> split -> id maps -> CSR equals the reference's DataSetLoader / Splitter / DataSet on the same file
    assert all(r["data_plane"].values()), r["data_plane"]


def test_get_single_recommendation_has_the_reference_signature():
    from elliot_amd.recommender.recommender_utils_mixin import RecMixin
    assert list(inspect.signature(RecMixin.get_single_recommendation).parameters) == ["self", "mask", "k", "predictions", "offset", "offset_stop"]
