# the row-major twiddle table of the first pass of a 2^24-point transform (2^24 entries, 512 MiB) vs the two-table lookup
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for TW in 23 24; do
  echo "== STARKHIP_TW2_MAX_LOG=$TW (round $rep)"
  STARKHIP_TW2_MAX_LOG=$TW timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 2 && STARKHIP_TW2_MAX_LOG=$TW timeout -k 10 100 python3 tools/ntt_batch_time.py 25 1 || exit 1
done; done
