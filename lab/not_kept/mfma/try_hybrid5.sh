# config 5, FRI commits and the bench's secondary legs under the hybrid path vs the VALU passes, one session
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for P in valu hybrid; do export STARKHIP_NTT_PATH=$P
  echo "== path $P (round $rep)"
  timeout -k 10 200 python3 bench.py --workload c5 --no-cpu-baseline --no-extras | grep -o '"value": [0-9.]*' && timeout -k 10 200 python3 tools/fri_profile.py 20:1 14:1 16:32 || exit 1
done; done
