#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace: for the LAST `count` dispatches, kernel time vs idle gaps on the queue.
usage: trace_gaps.py <kernel_trace.csv> <count>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cnt = int(sys.argv[2])
last = rows[-cnt:]
t0 = int(last[0]["Start_Timestamp"])
busy = 0
prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    busy += (e - s)
    print("%8.1f us  +%5.1f gap  %6.1f us  grid %s x %s  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"],
                                                            r["Kernel_Name"][:70]))
    prev_end = e
wall = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
print("wall %.1f us, kernels %.1f us, idle %.1f us over %d dispatches" % (wall, busy / 1e3, wall - busy / 1e3, cnt))
