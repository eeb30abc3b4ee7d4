#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs per (kernel, grid size): mean counters and duration of the LARGEST grid of every kernel
(a kernel name covers many sizes in a FRI commit; the averages over all of them say nothing).  Several run directories
(one per counter set, same command) are merged by (kernel, grid).  Usage: pmc_by_grid.py DIR... [--all-grids]"""
import collections, csv, glob, os, sys

def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]

def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    every = "--all-grids" in sys.argv
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    meta = {}
    for root in dirs:
        for cc in glob.glob(root + "/*/*_counter_collection.csv"):
            seen = set()
            for r in csv.DictReader(open(cc)):
                key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
                cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[key] = (r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    biggest = {}
    for (k, g) in cnt:
        if k not in biggest or g > biggest[k]:
            biggest[k] = g
    for (k, g) in sorted(cnt, key=lambda kg: (kg[0], -kg[1])):
        if not every and g != biggest[k]:
            continue
        d = dur[(k, g)]
        m = meta[(k, g)]
        print("%s  grid=%d wg=%s lds=%s vgpr=%s agpr=%s sgpr=%s scratch=%s  dispatches=%d  avg_us(under counters)=%.1f" %
              ((k, g) + m + (len(d), sum(d) / len(d))))
        c = {n: sum(v) / len(v) for n, v in cnt[(k, g)].items()}
        for n in sorted(c):
            print("     %-24s %.5g" % (n, c[n]))
        # derived
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            wc = c["SQ_WAVE_CYCLES"]
            print("     -> of wave-cycles: waiting(s_waitcnt/barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %% (VALU %.1f %%, LDS %.1f %%, VMEM %.1f %%)" % (
                100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VMEM", 0) / wc))
        if "SQ_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
            pass
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            print("     -> HBM-side bytes per dispatch: read 2 x FETCH_SIZE KiB = %.4g MB, written WRITE_SIZE KiB = %.4g MB" % (
                2 * c.get("FETCH_SIZE", 0) * 1024 / 1e6, c.get("WRITE_SIZE", 0) * 1024 / 1e6))
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            print("     -> VALU wave-instructions %.4g; per SIMD (1024 SIMDs) %.4g in %.4g cycles = %.2f cycles per VALU instruction per SIMD" % (
                c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU"] / 1024, cyc, cyc / (c["SQ_INSTS_VALU"] / 1024)))

if __name__ == "__main__":
    main()
