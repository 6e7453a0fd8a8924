#!/bin/bash
# gpurun_out/prof_<tag>/ (written by tools/profile_bench.sh on the GPU box) -> the tracked summaries under profiles/.
# usage: tools/collect_profiles.sh <tag> <prefix>      e.g. r04f r04_z
set -eu
TAG=$1; PRE=$2; SRC=gpurun_out/prof_$TAG
python tools/fmt_kernel_stats.py $SRC/kernel_stats_whole_bench.csv profiles/${PRE}_kernel_stats_whole_bench.txt
for D in $SRC/*_envs*; do
  W=$(basename $D)
  python tools/fmt_kernel_stats.py $D/kernel_stats_full_depth.csv profiles/${PRE}_${W}_kernel_stats_full_depth.txt
  cp $D/pmc_fetch.txt profiles/${PRE}_${W}_pmc_fetch_size.txt
  cp $D/pmc_write.txt profiles/${PRE}_${W}_pmc_write_size.txt
  cp $D/pmc_mfma_busy.txt profiles/${PRE}_${W}_pmc_mfma_busy.txt
done
cp $SRC/pmc_traffic.json profiles/pmc_traffic.json
