#!/usr/bin/env python
"""Timeline of one RK stage from a rocprofv3 --kernel-trace run (csv): every kernel with its queue (stream), start and
duration, and how much of the time of the communication-stream kernels (RCCL send/recv) runs under kernels of the
compute stream.  usage: timeline.py <dir> [stage-index]"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                     r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    # stages: from one k_sweep12s to the next
    starts = [i for i, r in enumerate(rows) if "k_sweep12s" in r[2]]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts)//2
    a, b = starts[which], starts[which + 1]
    seg = rows[a:b]
    t0 = seg[0][0]
    print("# stage %d of %d: %d kernels, %.1f us from the first start to the next stage's first start" % (
        which, len(starts), len(seg), (rows[b][0] - t0)/1e3))
    print("%10s %10s  %-6s %s" % ("start_us", "dur_us", "queue", "kernel"))
    for s, e, n, q in seg:
        print("%10.1f %10.1f  %-6s %s" % ((s - t0)/1e3, (e - s)/1e3, q, n[:90]))
    # overlap of the communication kernels with compute kernels, over ALL stages
    comm = [r for r in rows if "nccl" in r[2].lower() or "rccl" in r[2].lower()]
    comp = [r for r in rows if r[2].startswith("akmi::")]
    tot = ov = 0
    j = 0
    for s, e, n, q in comm:
        tot += e - s
        for cs, ce, cn, cq in comp:
            if ce <= s:
                continue
            if cs >= e:
                break
            if cq != q:
                ov += max(0, min(e, ce) - max(s, cs))
    if comm:
        print("# RCCL kernels: %d launches, %.1f us each on average; %.0f %% of their time under a compute-stream kernel" % (
            len(comm), tot/len(comm)/1e3, 100.0*ov/max(tot, 1)))


if __name__ == "__main__":
    main()
