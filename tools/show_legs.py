"""The legs of a bench.py JSON line on one row (value, value_at_target_depth, latency by exit, batched, batched_groups).  usage: show_legs.py bench.json"""
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], "value", d["value"], "target", d["value_at_target_depth"]["value"], "lat", d["latency_ms_by_exit"], "batched", (d.get("batched") or {}).get("value"), "groups", (d.get("batched_groups") or {}).get("value"))
