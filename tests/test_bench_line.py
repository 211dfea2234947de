"""The stdout line of bench.py stays one a consumer with a bounded read can parse (round 4's 22 KB line came back `parsed: null` from
the driver): compact_line() applied to the fattest report on record -- the committed round-4 line with every leg -- is strict JSON,
below the limit, and still carries the contract's fields, `roofline` and `cpu_baseline`.  (CPU; the GPU contract test checks the same
on a live run.)"""
import importlib.util
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_line_test", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _strict(text):
    def bad(c):
        raise ValueError(c)
    return json.loads(text, parse_constant=bad)


def test_compact_line_of_the_fattest_report_is_small_strict_and_complete():
    b = _bench()
    full = json.load(open(os.path.join(REPO, "profiles", "r04_n_bench_line.json")))
    assert len(json.dumps(full)) > 20000                       # the report that broke the consumer
    full["roofline"]["traffic"] = 5.14e9                       # + the stamps a traffic file adds
    full["topk"]["roofline"]["traffic"] = float("nan")         # + a non-finite float somewhere
    text = json.dumps(b.compact_line(full), allow_nan=False, separators=(",", ":"))
    assert len(text.encode()) <= b.LINE_LIMIT <= 6144, len(text)
    d = _strict(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "topk_users_per_s", "topk_ms_per_block", "topk_frac"):
        assert key in d, key
    assert "10M users x 1M items" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert set(("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and r["traffic"] == 5.14e9 and "builder" in d["traffic_source"]
    assert d["topk"]["roofline"]["traffic"] is None            # NaN -> null
    assert not any(k.endswith("GBs") and isinstance(v, float) and v > r["peak"] for k, v in r.items())
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 16 and cb["value"] > 0 and cb["topk"]["value"] > 0 and cb["sample"]
    assert set(d["legs"]) == {"c2", "c5_per_gpu", "batch_sweep", "plugin_e2e", "vae", "neumf"}
    assert d["legs"]["vae"]["roofline"]["kernel"] == "k_gemm_b3" and d["legs"]["neumf"]["topk"]["roofline"]["kernel"] == "k_nmf_screen"


def test_emit_drops_summaries_rather_than_exceed_the_limit(tmp_path, capsys):
    b = _bench()
    full = json.load(open(os.path.join(REPO, "profiles", "r04_n_bench_line.json")))
    full["batch_sweep"]["points"] = full["batch_sweep"]["points"] * 40           # a leg that would blow the line up
    args = type("A", (), {"legs_file": str(tmp_path / "legs.json")})()
    b.emit(full, args)
    out = capsys.readouterr().out.rstrip().splitlines()
    assert len(out) == 1 and len(out[0].encode()) <= b.LINE_LIMIT
    d = _strict(out[0])
    assert "legs" not in d and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    kept = _strict(open(tmp_path / "legs.json").read())
    assert len(kept["batch_sweep"]["points"]) == 240                             # the side file keeps everything
