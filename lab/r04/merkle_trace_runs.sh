cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04mt2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/merkle_time.py > $O/log.txt 2>&1 || { tail $O/log.txt; exit 1; }
grep merkelize $O/log.txt
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/r04mt2/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
big = [r for r in rows if int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) >= 1048576 and "merkle" in r["Kernel_Name"]]
t0 = int(big[0]["Start_Timestamp"])
for r in big:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%10.1f us  %8.1f us  grid %8s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size")), k))
P
