cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for V in "X=0" "STARKHIP_NTT_RADICES=8,9" "STARKHIP_NTT_RADICES=6,6,5" "STARKHIP_NTT_RADICES=7,5,5" "STARKHIP_NTT_RADICES=8,5,4" "STARKHIP_NTT_RADICES=5,6,6"; do
  echo "== [$V] (round $rep)"
  env $V timeout -k 10 200 python3 tools/fri_profile.py 14:1 | grep "steps" || exit 1
done; done
