# 2^24 single vector: alternative pass decompositions (STARKHIP_NTT_RADICES), forward + inverse
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for R in "8,8,8" "6,9,9" "9,9,6" "7,8,9" "9,8,7" "8,9,7" "7,9,8" "10,7,7" "7,7,10" "6,8,10" "10,8,6" "5,9,10" "4,10,10" "10,10,4" "6,6,6,6" "9,6,9"; do
  echo "== radices $R"; STARKHIP_NTT_RADICES=$R timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 || exit 1
done
