# NTT passes of the 2^24 transform with FEWER resident workgroups (diagnostic builds -DSHK_LDS_PAD_KIB=10 / 20: the dynamic LDS allocation of every
# tile pass is padded; 1024-element tiles 32 -> 42 / 52 KiB = 3 workgroups per CU instead of 4, 2048-element tiles 64 -> 74 KiB = still 2, 84 KiB = 1):
# how much does the pass lose per resident wave it gives up?  (profiles/r04_ntt_occupancy_sensitivity.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for L in "" _pad10 _pad20; do
  echo "== lib$L (round $rep)"
  export STARKHIP_LIB=$PWD/starks_amd/libstarkhip$L.so
  O=gpurun_out/occ$L; rm -rf $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --steps 20 --warmup 3 > $O.log 2>&1
  grep -o '"ms_per_step": [0-9.e+-]*' $O.log | head -1
  python3 - $O <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "ntt_pass" in r["Name"]:
        print("   %-50s avg %8.1f us  min %8.1f" % (r["Name"][:50], float(r["AverageNs"]) / 1e3, int(r["MinNs"]) / 1e3))
P
done; done
