# tile size per pass (STARKHIP_TILE_LOGS), the shapes of the bench and of config 5; "0,0,0" = the default rules
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== 2^$1 x $2, tiles $3"; STARKHIP_TILE_LOGS=$3 timeout -k 10 100 python3 tools/ntt_batch_time.py $1 $2 || exit 1; }
for rep in 1 2; do
for T in 0,0,0 11,11,10 11,10,11 11,11,11 11,9,10 11,10,9 12,10,10; do run 24 1 $T; done
for T in 0,0 12,11 11,12 10,11 11,10; do run 20 8 $T; done
for T in 0,0,0 11,10,10 10,11,10 10,10,11 11,11,10; do run 23 1 $T; done
for T in 0,0 10,11 11,10 12,11; do run 19 64 $T; done
for T in 0,0 11,10 10,11 9,10 10,9 11,11; do run 16 64 $T; done
done
