#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun).  Kernel trace and each PMC counter in separate passes, one set per
# WORKLOAD the bench quotes a roofline on (never paste one shape's traffic into another's line).
# usage: tools/profile_bench.sh <tag> [workloads...]   workloads = "<bench workload>/envs<B>", default: deer_b/envs1 deer_b/envs8 deer_9b/envs1
#   -> gpurun_out/prof_<tag>/<workload>/{trace_full_depth,fetch,write,mfma}, summaries *.txt, and gpurun_out/prof_<tag>/pmc_traffic.json
set -u
TAG=${1:-r03}
shift || true
WLS=${@:-"deer_b/envs1 deer_b/envs8 deer_9b/envs1"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# whole bench under the kernel trace (every leg)
BENCH="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --batched-envs 0 --surface-steps 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_whole_bench.csv
ARGS=""
for WL in $WLS; do
  W=${WL%/*}; B=${WL#*/envs}; D=$OUT/${W}_envs$B
  mkdir -p $D
  SMALL="python $ROOT/bench.py --workload $W --envs-per-gpu $B --full-depth-only 12 --no-cpu-baseline"
  # kernel trace of the workload the roofline is quoted on (full-depth steps only): avg launch duration per kernel recomputable from profiles/
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace_full_depth -- $SMALL > $D/trace.json 2> $D/trace.err
  find $D/trace_full_depth -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $D/kernel_stats_full_depth.csv
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- $SMALL > $D/fetch.json 2> $D/fetch.err
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- $SMALL > $D/write.json 2> $D/write.err
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $D/mfma -- $SMALL > $D/mfma.json 2> $D/mfma.err
  python $ROOT/tools/pmc_summary.py $D/fetch $D/pmc_fetch.txt > /dev/null
  python $ROOT/tools/pmc_summary.py $D/write $D/pmc_write.txt > /dev/null
  python $ROOT/tools/pmc_summary.py --mfma $D/mfma $D/pmc_mfma_busy.txt > /dev/null
  ARGS="$ARGS $WL $D/fetch $D/write"
done
python $ROOT/tools/pmc_summary.py --traffic $OUT/pmc_traffic.json $ARGS > /dev/null
# raw per-dispatch tables are large: keep only the summaries
find $OUT -name "*.csv" ! -name "kernel_stats*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
ls -la $OUT $OUT/*/
