# the LDS-resident matrix-core tile (STARKHIP_NTT_PATH=mfma_lds) against the VALU passes and the register-tile MFMA passes:
# parity first, then the timings of the three in one session
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
STARKHIP_NTT_PATH=mfma_lds timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or lde or fri_proofs_golden or stark_proofs_golden or rare_carry or random_plan" > gpurun_out/lds_parity.log 2>&1 || { tail -30 gpurun_out/lds_parity.log; echo PARITY_FAILED; exit 1; }
tail -1 gpurun_out/lds_parity.log
for rep in 1 2; do for P in valu mfma mfma_lds; do
  export STARKHIP_NTT_PATH=$P
  echo "== path $P (round $rep)"
  timeout -k 10 100 python3 tools/ntt_batch_time.py 20 1 8 32 && timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 && timeout -k 10 100 python3 tools/ntt_batch_time.py 21 1 8 || exit 1
done; done
