#!/usr/bin/env python3
"""rocprofv3 kernel_stats.csv -> fixed-width table (profiles/*.txt).  usage: fmt_kernel_stats.py <csv> <out.txt> [header-file]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
out = [f"{'kernel':100s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'pct':>6s}"]
for r in rows:
    out.append(f"{r['Name'][:100]:100s} {int(r['Calls']):8d} {int(r['TotalDurationNs']) / 1e6:10.2f} {float(r['AverageNs']) / 1e3:9.2f} "
               f"{int(r['MinNs']) / 1e3:8.2f} {int(r['MaxNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}")
hdr = open(sys.argv[3]).read() if len(sys.argv) > 3 else ""
open(sys.argv[2], "w").write(hdr + "\n".join(out) + "\n")
