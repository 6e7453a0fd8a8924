#!/bin/bash
# Effective shader clock under the frame-tile GEMM from the counters: GRBM_GUI_ACTIVE / (End - Start) per dispatch, for the c_fc frame tile at
# 4 frames (64 workgroups) and 16 frames (256 workgroups), together with SQ_BUSY_CYCLES / SQ_WAVE_CYCLES of the same dispatches.
#   usage: tools/dvfs_pmc.sh   -> gpurun_out/dvfs_pmc/summary.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/dvfs_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for F in 4 16; do
  M=$((257 * F))
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/grbm_$F -- python $ROOT/tools/gemm_one.py $M 4096 1024 63 40 > /dev/null 2> $OUT/grbm_$F.err
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq_$F -- python $ROOT/tools/gemm_one.py $M 4096 1024 63 40 > /dev/null 2> $OUT/sq_$F.err
done
python - <<PY > $OUT/summary.txt
import csv, glob, collections
for F in (4, 16):
    rows = collections.defaultdict(dict)
    for tag in ("grbm", "sq"):
        for f in glob.glob(f"$OUT/{tag}_{F}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_" not in r["Kernel_Name"]:
                    continue
                key = (tag, r["Dispatch_Id"])
                rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
                if "Start_Timestamp" in r:
                    rows[key]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    g = [v for (t, _), v in rows.items() if t == "grbm" and "GRBM_GUI_ACTIVE" in v and v.get("ns", 0) > 0]
    s = [v for (t, _), v in rows.items() if t == "sq"]
    print(f"c_fc frame tile, {F} frames ({F * 16} workgroups): {len(g)} dispatches")
    if g:
        g = g[len(g) // 4:]                                     # skip the ramp-up dispatches
        ns = sum(v["ns"] for v in g) / len(g)
        ga = sum(v["GRBM_GUI_ACTIVE"] for v in g) / len(g)
        gc = sum(v.get("GRBM_COUNT", 0) for v in g) / len(g)
        print(f"  duration {ns / 1e3:8.2f} us   GRBM_GUI_ACTIVE {ga:12.0f}   GRBM_COUNT {gc:12.0f}")
        for div in (1, 8):
            print(f"  GRBM_GUI_ACTIVE / {div} / duration = {ga / div / ns:6.3f} GHz")
    if s:
        s = s[len(s) // 4:]
        for k in sorted(s[0]):
            print(f"  {k:28s} {sum(v.get(k, 0) for v in s) / len(s):16.1f}")
PY
cat $OUT/summary.txt
find $OUT -name "*.db" -delete
