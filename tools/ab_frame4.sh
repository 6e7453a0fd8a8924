for v in 0 1 2 0 1; do
  DEER_GEMM_FRAME4=$v python bench.py --steps 60 --warmup 10 --scripted-steps 30 --on-policy-steps 30 --surface-steps 30 --latency-reps 2 --no-cpu-baseline > gpurun_out/ab_f4_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_f4_$v.json"))
w=d["window"]
print("FRAME4=$v value %.1f batched %.1f max_envs %.1f groups %.1f window %s hidden_only %.3f" % (d["value"], d["batched"]["value"], d["batched_max_envs"]["value"], d["batched_groups"]["value"], w["ms_per_window_by_frames_per_group"], w["hidden_states_only"]["ms_per_window"]))
PY
done
