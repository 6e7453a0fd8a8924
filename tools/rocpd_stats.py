#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace as a --stats style table.
usage: rocpd_stats.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"{'kernel':96s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'pct':>6s}"]
    for r in rows:
        lines.append(f"{r[0][:96]:96s} {r[1]:8d} {r[2] / 1e3:10.2f} {r[3]:9.2f} {r[4]:8.2f} {r[5]:8.2f} {100 * r[2] / tot:6.2f}")
    lines.append(f"TOTAL kernel time {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} dispatches")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
