#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun).  Kernel trace and each PMC counter in separate passes.
# usage: tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/{trace,fetch,write}/..., summaries in gpurun_out/prof_<tag>/*.txt
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --batched-envs 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
SMALL="python $ROOT/bench.py --full-depth-only 12 --no-cpu-baseline"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $SMALL > $OUT/fetch_bench.json 2> $OUT/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $SMALL > $OUT/write_bench.json 2> $OUT/write.err
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- $SMALL > $OUT/mfma_bench.json 2> $OUT/mfma.err
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python $ROOT/tools/pmc_summary.py $OUT/fetch $OUT/pmc_fetch.txt > /dev/null
python $ROOT/tools/pmc_summary.py $OUT/write $OUT/pmc_write.txt > /dev/null
python $ROOT/tools/pmc_summary.py --traffic $OUT/fetch $OUT/write $OUT/pmc_traffic.json > /dev/null
python $ROOT/tools/pmc_summary.py --mfma $OUT/mfma $OUT/pmc_mfma_busy.txt > /dev/null
# raw per-dispatch tables are large: keep only the summaries
find $OUT -name "*.csv" ! -name "kernel_stats.csv" -size +2M -delete
find $OUT -name "*.db" -delete
ls -la $OUT
head -25 $OUT/kernel_stats.csv
head -12 $OUT/pmc_fetch.txt
