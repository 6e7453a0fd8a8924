#!/bin/bash
# PMC passes (separate rocprofv3 --pmc runs, per the guide) over ONE python command, summarised for the kernels whose name contains a pattern.
# usage: tools/kernel_pmc.sh <tag> <kernel name substring> <python script + args ...>   -> gpurun_out/pmc_<tag>/summary.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; PAT=$2; shift 2
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python $ROOT/"$@" > $OUT/p$i.out 2> $OUT/p$i.err
done
python - <<PY > $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
names = set()
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$PAT" in r["Kernel_Name"]:
            names.add(r["Kernel_Name"][:100])
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("kernels:", sorted(names))
for k, (t, n) in sorted(agg.items()):
    print(f"{k:40s} {t / n:16.1f}  (per dispatch, {n} dispatches)")
PY
cat $OUT/summary.txt
tail -2 $OUT/p1.out
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +1M -delete
