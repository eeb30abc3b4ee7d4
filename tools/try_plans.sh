# experiment: two-pass NTT plans (STARKHIP_NTT_RADICES / STARKHIP_TILE_LOG_BIG / STARKHIP_XCD_SWZ) -- forward+inverse timing
set -e
run() {  # radices logn batches...
  local rad=$1 logn=$2; shift 2
  echo "== radices $rad"
  STARKHIP_XCD_SWZ=1 STARKHIP_NTT_RADICES=$rad timeout -k 10 120 python tools/ntt_batch_time.py $logn "$@"
}
run 6,6,5 17 1 64 256
run 8,9 17 1 64 256
run 9,8 17 1 64 256
run 6,6,6 18 1 32 128
run 9,9 18 1 32 128
run 7,6,6 19 1 16 64
run 9,10 19 1 16 64
run 10,9 19 1 16 64
run 7,7,6 20 1 8 32
run 10,10 20 1 8 32
run 7,7,7 21 1 4 16
run 10,11 21 1 4 16
run 11,10 21 1 4 16
