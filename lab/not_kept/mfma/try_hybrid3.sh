# hybrid tile pass with 1024-element tiles (four workgroups per CU) and odd radices (two shared-twiddle groups per tile)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export STARKHIP_NTT_PATH=hybrid STARKHIP_HYBRID_TILE_LOG=10
for R in 9,9,6 7,7,5,5; do
STARKHIP_NTT_RADICES=$R timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_ntt_golden_vectors or test_ntt_every_size_vs_oracle or test_ntt_padding_and_batch or test_ntt_large_digests_vs_oracle_fixture or test_lde_golden" > gpurun_out/hy3_parity.log 2>&1 || { tail -30 gpurun_out/hy3_parity.log; echo PARITY_FAILED; exit 1; }
tail -1 gpurun_out/hy3_parity.log
done
run() { echo "== path=$STARKHIP_NTT_PATH tile_log=$STARKHIP_HYBRID_TILE_LOG radices=$STARKHIP_NTT_RADICES: $*"; timeout -k 10 100 python3 tools/ntt_batch_time.py $* || exit 1; }
for P in valu hybrid; do export STARKHIP_NTT_PATH=$P
  for R in 9,9,6 8,8,8 9,8,7 7,9,8; do STARKHIP_NTT_RADICES=$R run 24 1; done
  for R in 9,9 ; do STARKHIP_NTT_RADICES=$R run 18 8 32; done
  for R in 9,10 9,5,5 7,7,5; do STARKHIP_NTT_RADICES=$R run 19 64; done
  for R in 10,10 9,9,2 7,7,6 5,5,5,5; do STARKHIP_NTT_RADICES=$R run 20 8; done
done
