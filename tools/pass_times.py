#!/usr/bin/env python3
"""durations of the last six NTT pass launches (one forward + one inverse transform) of a rocprofv3 kernel trace"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "ntt_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-6:]:
    print("   %-44s %8.1f us" % (r["Kernel_Name"].split("(")[0][-44:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
