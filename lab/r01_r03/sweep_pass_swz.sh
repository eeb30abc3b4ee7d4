# XCD mapping mode per pass (STARKHIP_XCD_SWZS; 0 = plain order, 1 = XCD-local only for tiles narrower than 128 B (default), 2 = always)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== 2^$1 x $2, swizzle per pass $3"; STARKHIP_XCD_SWZS=$3 timeout -k 10 100 python3 tools/ntt_batch_time.py $1 $2 || exit 1; }
for rep in 1 2; do
for T in 1,1,1 2,1,1 1,2,1 1,1,2 2,2,1 0,1,1; do run 24 1 $T; done
for T in 1,1 0,1 1,0 2,1 1,2 0,0; do run 20 8 $T; done
for T in 1,1 0,1 1,0 2,2; do run 19 64 $T; done
done
