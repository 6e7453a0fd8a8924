"""One-line summary + the top kernel classes of a bench.py JSON line (stdin or file).  usage: show_bench.py [bench.json]"""
import json,sys
d=json.loads(open(sys.argv[1]).read() if len(sys.argv)>1 else sys.stdin.read())
r=d["roofline"]
print("B=%d value=%.1f steps/s ms_per_step=%.3f avg_exit=%.2f fulldepth_gpu_us=%.0f"%(d["config"]["envs_per_gpu"],d["value"],d["ms_per_step"],d["avg_exit_layer"],r["gpu_us_per_full_depth_step"]))
for k,v in list(r["classes"].items())[:9]: print("   ",k,v)
