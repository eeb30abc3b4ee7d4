# kernel-by-kernel timeline of ONE whole STARK proof (args: logsteps): start offset, duration and gap to the previous kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=${1:-14}
O=gpurun_out/r04st_$L; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/stark_time.py $L:1 > $O/log.txt 2>&1 || { tail $O/log.txt; exit 1; }
grep steps $O/log.txt
python3 - $O <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "fri_gather_all" in r["Kernel_Name"])
first = max(i for i, r in enumerate(rows[:last]) if "stark_interp" in r["Kernel_Name"])
seg = rows[first:last + 1]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = t0
tot_k = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%9.1f us  +%5.1f gap  %7.1f us  grid %8s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), k))
    prev_end = e; tot_k += e - s
print("launches %d, kernel time %.1f us, span %.1f us" % (len(seg), tot_k / 1e3, (prev_end - t0) / 1e3))
P
