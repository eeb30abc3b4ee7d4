# A/B of two builds on config 5 and the FRI commits, alternating, one session: A = starks_amd/libstarkhip.so, B = libstarkhip_ab.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for L in A B; do
  if [ $L = B ]; then export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_ab.so; else unset STARKHIP_LIB; fi
  echo "== lib $L (round $rep)"
  timeout -k 10 200 python3 bench.py --workload c5 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"value": [0-9.]*' || exit 1
  timeout -k 10 200 python3 tools/fri_profile.py 16:32 20:1 | grep steps || exit 1
done; done
