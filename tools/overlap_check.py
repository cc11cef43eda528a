#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace csv and reports how much of the conv kernels' time overlaps another conv kernel
(do two streams really run side by side?).  python tools/overlap_check.py trace_kernel_trace.csv"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv3x3" in r["Kernel_Name"]]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
tot = sum(e - s for s, e, _, _ in iv)
ov = 0
for i in range(1, len(iv)):
    ov += max(0, min(iv[i - 1][1], iv[i][1]) - iv[i][0])
print("%d conv launches, %.3f ms summed, %.3f ms of it overlapping the previous launch (%.1f %%)" % (len(iv), tot / 1e6, ov / 1e6, 100.0 * ov / max(tot, 1)))
print("queues:", sorted(set(q for _, _, q, _ in iv)), "streams:", sorted(set(s for _, _, _, s in iv)))
for s, e, q, st in iv[200:212]:
    print("  start +%9.1f us  dur %7.1f us  queue %s stream %s" % ((s - iv[200][0]) / 1e3, (e - s) / 1e3, q, st))
