#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean counter values per dispatch."""
import csv, glob, os, sys, collections
def summarize(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
def durations(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return d
if __name__ == "__main__":
    for root in sys.argv[1:]:
        print("==", root)
        cc = max(glob.glob(root + "/*/*_counter_collection.csv"), key=os.path.getmtime)
        kt = max(glob.glob(root + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
        agg, dur = summarize(cc), durations(kt)
        for k, cs in agg.items():
            if not any(t in k for t in ("ntt_pass", "merkle", "fold", "stark", "sample", "gather", "lincomb")):
                continue
            ds = dur.get(k, [0])
            print(" %s  n=%d  avg_us=%.1f" % (k[:60], len(ds), sum(ds) / len(ds)))
            for c, v in sorted(cs.items()):
                print("     %-24s %.4g" % (c, sum(v) / len(v)))
