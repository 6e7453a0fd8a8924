#!/bin/bash
# per-kernel durations of the 8-environment full-depth step with the 16-wave frame tiles (DEER_GEMM_FRAME4=0) and the four-wave ones (=1)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_f4_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  DEER_GEMM_FRAME4=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$v -- python $ROOT/bench.py --workload deer_b --envs-per-gpu 8 --full-depth-only 12 --no-cpu-baseline > $OUT/t$v.json 2> $OUT/t$v.err
  find $OUT/t$v -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_frame4_$v.csv
  python $ROOT/tools/fmt_kernel_stats.py $OUT/kernel_stats_frame4_$v.csv $OUT/kernel_stats_frame4_$v.full.txt; head -16 $OUT/kernel_stats_frame4_$v.full.txt | cut -c1-64,100-170 > $OUT/kernel_stats_frame4_$v.txt
  cat $OUT/kernel_stats_frame4_$v.txt
done
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +1M -delete
