"""bench.py prints ONE JSON line with the fields the driver reads (run here on a reduced workload so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--users", "60000", "--items", "8000", "--factors", "64", "--batch", "65536",
           "--topk-block", "16384", "--steps", "3", "--warmup", "1", "--cpu-topk-users", "32", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def check_common(d, n_gpus=1):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    for r in (d["roofline"], d["topk"]["roofline"]):
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert r["achieved"] > 0 and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert "traffic" in r and "kernel" in r
    assert d["topk"]["value"] > 0 and d["topk"]["unit"] == "users/s"


def test_default_line_has_every_field():
    d = run_bench()
    check_common(d)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "pairs/s" and cb["sample"]
    assert cb["topk"]["value"] > 0 and cb["topk"]["unit"] == "users/s"
    assert d["value"] > 50 * cb["value"]                      # (sanity, not a claim: the roofline fraction is the quality number)


def test_one_rank_sharded_path_prints_the_same_contract():
    d = run_bench("--force-sharded", "--no-cpu-baseline")
    check_common(d)
    assert d["topk"]["sharding"]                              # (world 1: the line still says "single"; the N > 1 code ran through RCCL)
