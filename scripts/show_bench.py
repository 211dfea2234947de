"""One-line digest of bench.py's stdout line (the compact summary) or of the full per-leg report (bench_legs.json) on stdin."""
import json
import sys

for l in sys.stdin:
    l = l.strip()
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    tk = d["topk"]
    print(d["value"], d["ms_per_step"], tk.get("users_per_s", tk.get("value")), str(d["config"].get("parallelism", ""))[:120])
    k = d["roofline"].get("kernels_ms_per_step")          # (only the full report carries the per-kernel breakdown)
    print(json.dumps(k if k is not None else {x: d["roofline"].get(x) for x in ("kernel", "frac", "traffic")}))
