# the hybrid tile pass: its mapping and barriers with VALU butterflies (STARKHIP_HYBRID_MATH=valu), with the matrix-core blocks, and
# -- timing only, wrong results -- with NO arithmetic in the shared groups (=skip: the ceiling of any faster butterfly there)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# skip / frag0 exist in the diagnostic library only (make -C starks_amd/csrc stamps)
export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_stamps.so
for V in "valu x" "hybrid valu" "hybrid mfma" "hybrid frag0" "hybrid skip"; do set -- $V
  export STARKHIP_NTT_PATH=$1 STARKHIP_HYBRID_MATH=$2
  echo "== path $1 math $2"
  timeout -k 10 100 python3 tools/ntt_batch_time.py 20 1 8 32 && timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 && timeout -k 10 100 python3 tools/ntt_batch_time.py 19 64 || exit 1
done
