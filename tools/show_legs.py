import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], "value", d["value"], "target", d["value_at_target_depth"]["value"], "lat", d.get("latency",{}).get("ms_by_exit") or [k for k in d if "lat" in k], "batched", d["batched"]["value"], d["batched"]["avg_exit_layer"], "groups", d["batched_groups"]["value"], "window", d["window"]["hidden_states_only"]["ms_per_window"])
