# radix mixes of the LDS-resident matrix-core tile (radix above 2^7 falls to the VALU pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export STARKHIP_NTT_PATH=mfma_lds
for R in 7,7,6 6,7,7 7,6,7 6,6,8 5,5,10 5,5,5,5 6,6,4,4 5,7,8 10,10; do
  echo "== 2^20 radices $R"; STARKHIP_NTT_RADICES=$R timeout -k 10 100 python3 tools/ntt_batch_time.py 20 1 8 32 || exit 1
done
for R in 6,6,6,6 7,7,5,5 5,5,7,7 6,6,5,7 8,8,8 7,7,10 6,6,6,6; do
  echo "== 2^24 radices $R"; STARKHIP_NTT_RADICES=$R timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 || exit 1
done
export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_stamps.so
for d in 0 1; do echo "== 2^20 x 8 vectors (6,6,8), pass $d"; STARKHIP_NTT_RADICES=6,6,8 STARKHIP_STAMP_PASS=$d timeout -k 10 100 python3 tools/lds_phases.py 20 8 || exit 1; done
for d in 0; do echo "== 2^20 x 8 vectors (5,5,10), pass $d"; STARKHIP_NTT_RADICES=5,5,10 STARKHIP_STAMP_PASS=$d timeout -k 10 100 python3 tools/lds_phases.py 20 8 || exit 1; done
