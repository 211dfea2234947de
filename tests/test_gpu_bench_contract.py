"""bench.py prints ONE JSON line with the fields the driver reads (run here on a reduced workload so that it takes seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_LIMIT = 6144     # what the driver's consumer is known to read whole (rounds 1-3 parsed <= 19 KB lines from a file, the 22 KB one
#                       of round 4 came back `parsed: null`; the line is now a summary and stays far below either)


def _strict(text):
    def bad(c):
        raise ValueError(f"non-JSON constant {c}")
    return json.loads(text, parse_constant=bad)


def run_bench(*extra, env=None, both=False):
    """Runs bench.py; returns the FULL per-leg report (the side file) after checking the stdout line (one line, compact, strict JSON,
    consistent with the report).  both=True: (line, full)."""
    import tempfile
    legs = tempfile.NamedTemporaryFile(prefix="bench_legs_", suffix=".json", delete=False)
    legs.close()
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--users", "60000", "--items", "8000", "--factors", "64", "--batch", "65536",
           "--topk-block", "16384", "--steps", "3", "--warmup", "1", "--cpu-topk-users", "32", "--cpu-seconds", "0.5",
           "--legs-file", legs.name, *extra]
    e = dict(os.environ)
    e.update(env or {})
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO, env=e)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        assert out.stdout.rstrip().splitlines()[-1] == lines[0]            # the LAST thing on stdout
        assert len(lines[0].encode()) <= LINE_LIMIT, len(lines[0])
        line = _strict(lines[0])
        full = _strict(open(legs.name).read())
    finally:
        os.unlink(legs.name)
    check_line_against_report(line, full)
    return (line, full) if both else full


def check_line_against_report(line, full):
    """The stdout line carries the contract's fields and agrees with the full report to the 6 digits it keeps."""
    def close(a, b):
        return a == b or abs(a - b) <= 1e-5 * abs(b)
    for key in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[key] == full[key], key
    for key in ("value", "ms_per_step", "topk_users_per_s", "topk_ms_per_block", "topk_frac"):
        assert close(line[key], full[key]), key
    assert "workload" in line["config"] and "model" not in line["config"]
    for key in ("users", "items", "factors", "batch", "topk_block", "k", "world_size_observed"):
        assert line["config"][key] == full["config"][key]
    check_roofline(line["roofline"], tol=1e-4)
    check_roofline(line["topk"]["roofline"], tol=1e-4)
    assert line["roofline"]["kernel"] == full["roofline"]["kernel"] and close(line["roofline"]["frac"], full["roofline"]["frac"])
    assert all(not (k.endswith("GBs") and isinstance(v, float) and v > line["roofline"]["peak"]) for k, v in line["roofline"].items())
    if "cpu_baseline" in full:
        cb = line["cpu_baseline"]
        assert cb["kind"] == full["cpu_baseline"]["kind"] and cb["cores"] == full["cpu_baseline"]["cores"] and cb["sample"]
        assert close(cb["value"], full["cpu_baseline"]["value"]) and close(cb["topk"]["value"], full["cpu_baseline"]["topk"]["value"])
    for leg in ("c2", "c5_per_gpu", "vae", "neumf", "batch_sweep", "plugin_e2e"):
        assert (leg in full) == (leg in line.get("legs", {})), leg
    assert line["legs_file"]


def check_roofline(r, tol=1e-9):
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["achieved"] > 0 and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < tol
    assert "traffic" in r and "kernel" in r


def check_common(d, n_gpus=1):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == n_gpus and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["world_size_observed"] == n_gpus
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - n_gpus * d["config"]["batch"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    for r in (d["roofline"], d["topk"]["roofline"]):
        check_roofline(r)
    assert d["topk"]["value"] > 0 and d["topk"]["unit"] == "users/s"
    # the second half of the metric is also at the top level of the line (the driver's record keeps top-level scalars)
    assert d["topk_users_per_s"] == d["topk"]["value"] and d["topk_ms_per_block"] == d["topk"]["ms_per_step"]
    assert d["topk_frac"] == d["topk"]["roofline"]["frac"]


def test_default_line_has_every_field():
    d = run_bench("--legs", "bpr,metrics")
    check_common(d)
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and 1 <= cb["cores"] <= os.cpu_count() and cb["value"] > 0 and cb["unit"] == "pairs/s" and cb["sample"]
    assert cb["topk"]["value"] > 0 and cb["topk"]["unit"] == "users/s"
    assert cb["port_1core"]["cores"] == 1 and cb["port_1core"]["value"] > 0 and cb["port_1core"]["topk"]["value"] > 0
    assert d["metrics"]["value"] > 0
    fr = d["topk"]["fragile_users"]
    assert fr["users"] == d["config"]["topk_block"] and 0 <= fr["fragile"] <= fr["users"]


def test_pipelined_and_plain_training_step_lines():
    """Default: the next batch is drawn and sorted under the current step (roofline carries both the live and the back-to-back
    figure of the dominant kernel); --no-pipeline: the plain sequence."""
    d = run_bench("--legs", "bpr", "--no-cpu-baseline")
    check_common(d)
    r = d["roofline"]
    assert "pipelined" in r and r["frac_back_to_back"] > 0 and abs(r["frac_back_to_back"] - r["achieved_back_to_back"] / r["peak"]) < 1e-9
    p = run_bench("--legs", "bpr", "--no-cpu-baseline", "--no-pipeline")
    check_common(p)
    assert "pipelined" not in p["roofline"]
    assert set(p["roofline"]["kernels_ms_per_step"]) == set(r["kernels_ms_per_step"])     # the same kernels either way


def test_secondary_legs_carry_their_rooflines():
    """vae = BASELINE configs[2], neumf = configs[3] per-GPU shape; here at toy shapes (the default shapes run in bench.py itself)."""
    d = run_bench("--legs", "bpr,c2,metrics,vae,neumf", "--no-cpu-baseline", "--vae-shape", "3000,1500,96,32,256", "--neumf-shape", "5000,3000,32,8192",
                  "--c2-shape", "30000,4000")
    check_common(d)
    c2 = d["c2"]                                           # BASELINE configs[1] beside the headline (here small)
    assert c2["value"] > 0 and c2["unit"] == "pairs/s" and c2["topk"]["value"] > 0 and "30000 users x 4000 items" in c2["workload"]
    assert c2["metrics"]["value"] > 0
    for r in (c2["roofline"], c2["topk"]["roofline"]):
        check_roofline(r)
    for leg, unit in (("vae", "users/s"), ("neumf", "samples/s")):
        assert d[leg]["value"] > 0 and d[leg]["unit"] == unit and d[leg]["workload"]
        check_roofline(d[leg]["roofline"])
        assert d[leg]["roofline"]["bound"] == "mfma" and d[leg]["roofline"]["kernel"] in ("k_gemm_f32", "k_gemm_b3")
    tk = d["neumf"]["topk"]                                 # (3000 items: below the screened route's floor, the fp32 kernel alone)
    check_roofline(tk["roofline"])
    assert tk["value"] > 0 and tk["unscreened"]["ms_per_step"] > 0 and tk["screen"]["used"] is False and tk["roofline"]["kernel"] == "k_nmf_score"


def test_sweep_plugin_and_c5_legs():
    """batch_sweep (B x optimiser grid through el_bprmf_train_loop) and plugin_e2e (external.BPRMF_batch through RecMixin.train() +
    evaluate()) on the c2 leg's data, c5_per_gpu (BASELINE configs[4] per-GPU shape; here small, d = 256) -- every leg reports its
    repeats."""
    d = run_bench("--legs", "bpr,sweep,plugin,c5", "--no-cpu-baseline", "--c5-shape", "200000,30000,256", "--repeats", "2", "--c2-shape", "60000,8000")
    check_common(d)
    assert d["repeats"] == 2 and len(d["repeats_ms_per_step"]) == 2 and min(d["repeats_ms_per_step"]) <= d["ms_per_step"] <= max(d["repeats_ms_per_step"])
    pts = d["batch_sweep"]["points"]
    assert {(p["optimizer"], p["batch"]) for p in pts} == {(o, b) for o in ("adam_tf_dense", "adam_lazy") for b in (4096, 65536, 1 << 20)}
    assert all(p["value"] > 0 and p["unit"] == "pairs/s" and len(p["repeats_ms_per_step"]) == 2 for p in pts)
    pl = d["plugin_e2e"]
    assert pl["train_epoch_s"] > 0 and pl["evaluate_s"] > 0 and pl["train_pairs_per_s"] > 0 and 0 <= pl["nDCG"] <= 1 and 0 <= pl["Recall"] <= 1
    c5 = d["c5_per_gpu"]
    assert c5["value"] > 0 and c5["topk"]["value"] > 0 and "200000 users x 30000 items" in c5["workload"] and "d=256" in c5["workload"]
    for r in (c5["roofline"], c5["topk"]["roofline"]):
        check_roofline(r)


def test_one_rank_sharded_path_prints_the_same_contract():
    d = run_bench("--force-sharded", "--no-cpu-baseline")
    check_common(d)
    assert d["topk"]["sharding"]                              # (world 1: the N > 1 code ran through RCCL)
    assert d["collectives"] and d["collectives"][0]["op"] == "all_reduce" and d["collectives"][0]["bytes"] > 0


def test_gpus_2_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` without torchrun in front: bench.py starts the ranks itself and rank 0 prints n_gpus == 2.  The
    dev box has ONE GPU, so both ranks share cuda:0 over gloo (EL_BENCH_SHARED_GPU=1): the control flow, not a measurement."""
    d = run_bench("--gpus", "2", "--no-cpu-baseline", env={"EL_BENCH_SHARED_GPU": "1"})
    check_common(d, n_gpus=2)
    assert d["config"]["backend"] == "gloo"
    assert d["collectives"][0]["op"] == "all_reduce" and d["collectives"][0]["ms"] > 0
    ex = d["collectives"][0]["expected_ms"]                   # wire model beside the measurement: 2 (G-1)/G S bytes per rank
    assert abs(ex["bytes_on_wire_per_rank"] - d["collectives"][0]["bytes"]) < 1 and ex["ring_one_link_ms"] >= ex["direct_all_links_ms"] > 0
    sec = d["item_shard"]                                     # north_star's partitioning as the second leg
    assert sec["value"] > 0 and sec["topk"]["value"] > 0 and sec["topk"]["scaling"] == "strong"
    ops_seen = {c["op"] for c in sec["collectives"]}
    assert "all_gather" in ops_seen and all(c["bytes"] > 0 and c["ms"] > 0 for c in sec["collectives"])
    check_roofline(sec["roofline"])


def test_refuses_more_gpus_than_the_node_has():
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "64"]
    e = dict(os.environ)
    e.pop("EL_BENCH_SHARED_GPU", None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=REPO, env=e)
    assert out.returncode != 0 and "refusing" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_collectives_through_the_c_abi_one_rank():
    """--comm abi: the sharded legs with RCCL called through el_comm_* (one rank: an API check on the 1-GPU box)."""
    d = run_bench("--force-sharded", "--no-cpu-baseline", "--comm", "abi")
    check_common(d)
    assert d["config"]["collectives_through"] == "abi"
    assert d["collectives"][0]["op"] == "all_reduce" and d["collectives"][0]["ms"] > 0


def test_deferred_legs_are_timed_in_their_steady_state():
    """A shape whose batches leave most rows alone (4 B <= U, 2 B <= I): the user AND the item table run their deferred decay, the line
    says so, and the timed region starts after cover batches gave every row a gradient (no row at the m = v = 0 fixed point)."""
    cmd_extra = ("--legs", "bpr", "--no-cpu-baseline", "--users", "300000", "--items", "150000", "--batch", "65536")
    d = run_bench(*cmd_extra)
    r = d["roofline"]
    assert "deferred_decay" in r and r["valu"]["steady_state"] and "item_side" in r and "deferred" in r["item_side"]
    assert r["valu"]["element_steps_per_step"] == (300000 + 150000) * 64
    assert 0 < r["item_rows_per_step"] <= 2 * 65536 and 0 < r["user_rows_per_step"] <= 65536
