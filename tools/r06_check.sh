#!/bin/bash
# round-6 validation on a GPU box: the whole GPU suite as the driver runs it, smoke(), the default bench line and the driver's --steps 20
# form.  "evidence" as first argument adds the two measurements DESIGN.md 4.2 cites (ViT N=1024 projections without split-K: in the step
# and stand-alone).
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r06_suite_g.txt 2>&1; echo "suite rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.txt 2>&1; echo "smoke rc=$?"
python bench.py > gpurun_out/r06_bench_e_fp16.json 2> gpurun_out/r06_bench_e.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_e_steps20.json 2>> gpurun_out/r06_bench_e.err; echo "bench20 rc=$?"
if [ "${1:-}" = "evidence" ]; then
  DEER_VIT_SPLIT=1,1,4,8 python bench.py --batched-envs 0 --surface-steps 0 --no-cpu-baseline > gpurun_out/r06_bench_d_vit_split_1_1.json 2>> gpurun_out/r06_bench_e.err; echo "split11 rc=$?"
  for f in 1 2; do timeout 300 python tools/bench_resadd_direct.py $f > gpurun_out/r06_resadd_direct_frames$f.txt 2>&1; echo "resadd_direct $f rc=$?"; done
fi
tail -3 gpurun_out/r06_suite_g.txt
