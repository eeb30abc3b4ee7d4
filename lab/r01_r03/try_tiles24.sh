# 2^24 single vector: tile size / XCD mapping variants of the VALU passes, with per-pass kernel times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for V in "STARKHIP_TILE_LOG=10" "STARKHIP_TILE_LOG=11" "STARKHIP_TILE_LOG=9" "STARKHIP_XCD_SWZ=2" "STARKHIP_XCD_SWZ=2 STARKHIP_TILE_LOG=11"; do
  echo "== $V"; env $V timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 || exit 1
done
