set -e
for v in 22 23 24; do
echo "== STARKHIP_TW2_MAX_LOG=$v"
STARKHIP_TW2_MAX_LOG=$v timeout -k 10 120 python tools/ntt_batch_time.py 24 1 2
STARKHIP_TW2_MAX_LOG=$v timeout -k 10 120 python tools/ntt_batch_time.py 23 1 4
STARKHIP_TW2_MAX_LOG=$v timeout -k 10 120 python tools/fri_profile.py 20:1
done
