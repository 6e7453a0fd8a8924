#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters from its CSV output (counter_collection.csv).
usage: pmc_summary.py <dir-or-csv> [out.txt]
       pmc_summary.py --traffic <out.json> <workload key> <fetch-dir> <write-dir> [...]    (per-class HBM bytes per launch, per workload, for bench.py)   (FETCH_SIZE/WRITE_SIZE are reported in KiB by rocprofv3; gfx950 FETCH_SIZE
under-reports wide coalesced reads by 2x - see /opt/skills/guides/MI355X_MICROARCH.md 'HBM' - the x2 column applies it)"""
import csv
import glob
import os
import sys
from collections import defaultdict


CLASS_OF = (("trunk_wide_gemm", "deer_trunk_wide_gemm"), ("trunk_mpt_attn", "deer_trunk_mpt_attn"), ("xattn_fused", "deer_xattn_fused"),
            ("gemm_tiled", "deer_gemm_bf16_nt"), ("gemm_frame", "deer_gemm_bf16_nt"), ("gemm_ring", "deer_gemm_bf16_nt"), ("gemm_skinny_hl", "deer_gemm_skinny_hl"),
            ("gemm_skinny", "deer_gemm_skinny"), ("slab_gelu_split", "deer_slab_gelu_split"), ("attn_mfma_kernel<false>", "deer_attn_mfma_hd64"),
            ("attn_mfma_kernel<true>", "deer_xattn_mfma"), ("resadd_ln", "deer_resadd_ln"), ("ln_rows", "deer_layernorm_rows"),
            ("head_lstm", "deer_head_lstm_layer"))


def per_class(src):
    import json
    files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                for key, cls in CLASS_OF:
                    if key in name:
                        agg[cls][0] += float(row.get("Counter_Value", 0) or 0)
                        agg[cls][1] += 1
                        break
    return agg


def traffic(out, triples):
    """triples: [workload key, fetch dir, write dir, ...] -> {"workloads": {key: {"kernel_source_hash", "classes": {...}}}}"""
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                                          # the stamp bench.py checks: hash of the kernel sources profiled
    stamp = bench.kernel_source_hash()
    wls = {}
    for i in range(0, len(triples), 3):
        key, fetch_dir, write_dir = triples[i:i + 3]
        f, w = per_class(fetch_dir), per_class(write_dir)
        classes = {}
        for cls in f:
            fk = f[cls][0] / max(f[cls][1], 1)                     # KiB per dispatch as reported
            wk = w[cls][0] / max(w[cls][1], 1) if cls in w else 0.0
            classes[cls] = {"fetch_kib_reported": round(fk, 1), "write_kib_reported": round(wk, 1), "dispatches": f[cls][1],
                            "hbm_bytes_per_launch": int(round((2.0 * fk + wk) * 1024))}
        wls[key] = {"kernel_source_hash": stamp, "classes": classes}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, full-depth steps of the named workload; FETCH_SIZE x2 "
                       "(gfx950 128-B requests tallied at 64 B), WRITE_SIZE as reported", "workloads": wls}, open(out, "w"), indent=1)
    print(json.dumps(wls, indent=1))


def mfma(src, out):
    """MFMA-busy % per kernel: SQ_VALU_MFMA_BUSY_CYCLES (busy cycles summed over the 1024 SIMDs) / (kernel-active cycles x
    1024 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so active cycles = GRBM_GUI_ACTIVE / 8."""
    files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(lambda: defaultdict(float))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row.get("Kernel_Name", "?")[:90], row.get("Dispatch_Id", "0"))
                per[k][row.get("Counter_Name", "?")] += float(row.get("Counter_Value", 0) or 0)
    agg = defaultdict(lambda: [0.0, 0.0, 0])
    for (name, _), c in per.items():
        act = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if act > 0:
            a = agg[name]
            a[0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            a[1] += act * 1024.0
            a[2] += 1
    lines = [f"{'kernel':90s} {'dispatches':>10s} {'active_us/dispatch':>18s} {'MFMA busy %':>12s}"]
    for name, (busy, cap, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{name:90s} {n:10d} {cap / 1024.0 / n / 2400.0:18.2f} {100.0 * busy / cap:12.2f}")
    txt = "\n".join(lines)
    print(txt)
    open(out, "w").write("# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); active_us assumes 2.4 GHz; PMC serialises dispatches\n"
                         + txt + "\n")


def main():
    if sys.argv[1] == "--traffic":
        return traffic(sys.argv[2], sys.argv[3:])
    if sys.argv[1] == "--mfma":
        return mfma(sys.argv[2], sys.argv[3])
    src = sys.argv[1]
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row.get("Kernel_Name", "?")[:80], row.get("Counter_Name", "?"))
                agg[k][0] += float(row.get("Counter_Value", 0) or 0)
                agg[k][1] += 1
    lines = [f"{'kernel':80s} {'counter':14s} {'dispatches':>10s} {'mean':>14s} {'mean_x2':>14s} {'total':>16s}"]
    for (k, c), (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        lines.append(f"{k:80s} {c:14s} {n:10d} {tot / n:14.2f} {2 * tot / n:14.2f} {tot:16.1f}")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
