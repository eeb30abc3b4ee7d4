# kernel trace of merkelize 2^24 (x3) with the tree's library: per-kernel durations by grid size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04mt; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/merkle_time.py > $O/log.txt 2>&1 || { tail $O/log.txt; exit 1; }
grep merkelize $O/log.txt
python3 - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/r04mt/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    d[(k, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(d.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print("%-40s grid %9d  n=%3d  min %8.1f us  avg %8.1f us" % (k, g, len(v), min(v), sum(v) / len(v)))
P
