#!/usr/bin/env python3
"""profiles/<round>_traffic.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of `bench.py` (tools/prof_traffic.sh).
HBM bytes per launch of the dominant kernel = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE
reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact."""
import csv, glob, json, os, sys, collections
def per_kernel(root, counter):
    f = max(glob.glob(root + "/*/*_counter_collection.csv"), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
def durations(root):
    f = max(glob.glob(root + "/*/*_kernel_stats.csv"), key=os.path.getmtime)
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(f))}
if __name__ == "__main__":
    fdir, wdir, sdir, out = sys.argv[1:5]
    vectors = int(sys.argv[5]) if len(sys.argv) > 5 else 8  # bench.py --batch of the profiled command
    logn = int(sys.argv[6]) if len(sys.argv) > 6 else 20
    F, W, D = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE"), durations(sdir)
    res = {"workload": "bench.py --logn %d --batch %d: 2^%d-point NTT + inverse NTT, %d vectors per step" % (logn, vectors, logn, vectors),
           "logn": logn, "vectors_per_step": vectors, "unit": "bytes per launch",
           "algorithmic_bytes_per_launch": None,  # 64 B per element per transform / passes, filled in below
           "correction": "2*FETCH_SIZE + WRITE_SIZE, counters in KiB (gfx950: FETCH_SIZE = 1/2 of a wide coalesced stream)",
           "kernels": {}}
    tot_b = tot_n = 0
    for k in F:
        if "ntt_pass" not in k and "ntt_ctile" not in k:
            continue
        fb = 2 * 1024 * sum(F[k]) / len(F[k])
        wb = 1024 * sum(W[k]) / len(W[k])
        calls, avg_ns = D.get(k, (len(F[k]), 0.0))
        res["kernels"][k] = {"launches": len(F[k]), "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                             "avg_duration_us_kernel_trace": avg_ns / 1e3}
        tot_b += (fb + wb) * len(F[k])
        tot_n += len(F[k])
    res["ntt_pass_kernel_mean_hbm_bytes_per_launch"] = tot_b / tot_n
    # launches per transform: the row pass (LAST = true) runs once per transform
    last = sum(v["launches"] for k, v in res["kernels"].items() if "true>" in k)
    res["passes_per_transform"] = round(tot_n / last) if last else None
    if last:
        res["algorithmic_bytes_per_launch"] = 64.0 * (1 << logn) * vectors / res["passes_per_transform"]
        res["traffic_over_algorithmic"] = res["ntt_pass_kernel_mean_hbm_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
    if len(sys.argv) > 7:  # directory of a --pmc SQ_INSTS_VALU ... run of the same command
        V = per_kernel(sys.argv[7], "SQ_INSTS_VALU")
        tot = 0.0
        for k in res["kernels"]:
            if k in V:
                res["kernels"][k]["valu_wave_instructions_per_launch"] = sum(V[k]) / len(V[k])
                tot += res["kernels"][k]["valu_wave_instructions_per_launch"] * res["kernels"][k]["launches"]
        if last and tot:
            # wave-instructions of all launches x 64 lanes / elements transformed (each transform = n * vectors elements)
            res["lane_instructions_per_element_per_transform"] = tot * 64.0 / (last * (1 << logn) * vectors)
            res["valu_source"] = "rocprofv3 --pmc SQ_INSTS_VALU (wave-instructions per dispatch), same command"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))
