import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["topk"]["value"], d["config"]["parallelism"][:120]); print(json.dumps(d["roofline"]["kernels_ms_per_step"]))
