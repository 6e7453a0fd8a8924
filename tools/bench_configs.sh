#!/bin/bash
# bench lines of the other BASELINE.json configurations on the current tree (fp16 arithmetic unless said): DeeR-S, exit_ratio 1.0, 9B, fp32
# usage (GPU box, repo root): bash tools/bench_configs.sh <prefix>   -> gpurun_out/<prefix>_bench_config_*.json
P=${1:-r06}
O="--no-cpu-baseline --batched-envs 0 --surface-steps 0 --no-two-groups --window-reps 0"
python bench.py --workload deer_s $O > gpurun_out/${P}_bench_config_deer_s.json 2>/dev/null; echo "deer_s rc=$?"
python bench.py --exit-ratio 1.0 $O > gpurun_out/${P}_bench_config_ratio10.json 2>/dev/null; echo "ratio10 rc=$?"
python bench.py --workload deer_9b $O > gpurun_out/${P}_bench_config_9b.json 2>/dev/null; echo "9b rc=$?"
python bench.py --precision fp32 $O > gpurun_out/${P}_bench_config_fp32.json 2>/dev/null; echo "fp32 rc=$?"
python - <<PY
import json
for n in ("deer_s","ratio10","9b","fp32"):
    try:
        d=json.load(open("gpurun_out/${P}_bench_config_%s.json" % n)); print(n, d["value"], d["avg_exit_layer"], d["dtype"], d["latency_ms_by_exit"])
    except Exception as e: print(n, "ERR", e)
PY
