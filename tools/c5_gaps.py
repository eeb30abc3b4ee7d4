#!/usr/bin/env python3
"""Queue timeline of the LAST batched launch (chunk) of a config-5 run from a rocprofv3 kernel trace: kernel time, idle time, and
every idle gap above 15 us with the kernels on either side.  usage: c5_gaps.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a chunk starts with the trace-point kernel's predecessor chain; use the last two stark_leaves launches as markers
marks = [i for i, r in enumerate(rows) if "stark_leaves_kernel" in r["Kernel_Name"]]
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
def busy_of(seg):
    return sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
# the chunk with the most kernel time (the run ends with single-unit checks, whose launches are small)
a, b = max(zip(marks, marks[1:]), key=lambda ab: busy_of(rows[ab[0]:ab[1]]))
seg = rows[a:b]
busy = busy_of(seg)
wall = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print("one chunk (from one packed-leaf launch to the next): %d dispatches, wall %.2f ms, kernels %.2f ms, idle %.2f ms" %
      (len(seg), wall / 1e6, busy / 1e6, (wall - busy) / 1e6))
for x, y in zip(seg, seg[1:] + [rows[b]]):
    gap = (int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3
    if gap > 15:
        print("  %7.1f us idle between %s and %s" % (gap, short(x["Kernel_Name"]), short(y["Kernel_Name"])))
import collections
acc = collections.Counter()
for r in seg:
    acc[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in acc.most_common(12):
    print("  %6.2f ms  %s" % (v / 1e6, k))
