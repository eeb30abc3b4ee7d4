cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
STARKHIP_NTT_PATH=hybrid timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_ntt_golden_vectors or test_ntt_every_size_vs_oracle or test_ntt_padding_and_batch or test_ntt_large_digests_vs_oracle_fixture" > gpurun_out/hy4_parity.log 2>&1 || { tail -30 gpurun_out/hy4_parity.log; echo PARITY_FAILED; exit 1; }
tail -1 gpurun_out/hy4_parity.log
for rep in 1 2; do for V in "valu x" "hybrid mfma" "mfma_lds x"; do set -- $V
  export STARKHIP_NTT_PATH=$1 STARKHIP_HYBRID_MATH=$2
  echo "== path $1 math $2"
  timeout -k 10 100 python3 tools/ntt_batch_time.py 20 1 8 32 && timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 && timeout -k 10 100 python3 tools/ntt_batch_time.py 19 64 || exit 1
done; done
