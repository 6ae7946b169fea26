#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (sqlite .db or *_kernel_trace.csv) as a small
text table: calls, total ms, average us, share.  Usage: kernel_stats.py <dir> [header...]"""
import csv
import glob
import os
import sqlite3
import sys


def rows_from(path):
    dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    out = {}
    if dbs:
        c = sqlite3.connect(dbs[0])
        for name, n, tot in c.execute("select name, count(*), sum(end-start) from kernels group by name"):
            out[name] = (n, tot)
        return out
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            n, t = out.get(r["Kernel_Name"], (0, 0))
            out[r["Kernel_Name"]] = (n + 1, t + d)
    return out


def main():
    rows = rows_from(sys.argv[1])
    tot = sum(t for _, t in rows.values()) or 1
    for h in sys.argv[2:]:
        print("# " + h)
    print("%-64s %8s %12s %12s %8s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for name, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        short = name.split("(")[0].replace("void ", "")
        print("%-64s %8d %12.3f %12.2f %7.2f%%" % (short[:64], n, t/1e6, t/1e3/n, 100.0*t/tot))


if __name__ == "__main__":
    main()
