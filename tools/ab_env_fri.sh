# A/B of environment knobs on the FRI commits / Merkle / STARK timings in ONE session: usage  bash tools/ab_env_fri.sh "A=1" "STARKHIP_X=1" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for V in "$@"; do
  echo "== [$V] (round $rep)"
  env $V timeout -k 10 200 python3 tools/fri_profile.py 14:1 16:1 16:32 20:1 | grep -i "steps\|ms" || exit 1
done; done
