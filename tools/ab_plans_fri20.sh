cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for V in "X=0" "STARKHIP_NTT_RADICES=8,7,8" "STARKHIP_NTT_RADICES=7,8,8" "STARKHIP_NTT_RADICES=9,7,7" "STARKHIP_NTT_RADICES=9,8,6" "STARKHIP_NTT_RADICES=8,9,6" "STARKHIP_NTT_RADICES=10,7,6" "STARKHIP_NTT_RADICES=9,9,5" "STARKHIP_NTT_RADICES=6,6,6,5"; do
  echo "== [$V] (round $rep)"
  env $V timeout -k 10 200 python3 tools/fri_profile.py 20:1 | grep "steps" || exit 1
done; done
