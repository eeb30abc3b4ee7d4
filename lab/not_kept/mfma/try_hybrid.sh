# the hybrid tile pass (STARKHIP_NTT_PATH=hybrid: VALU plans and tiles, lane-shared-twiddle groups on the matrix cores)
# against the VALU passes: parity first, then timings in one session
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
STARKHIP_NTT_PATH=hybrid timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or lde or fri_proofs_golden or stark_proofs_golden or rare_carry" > gpurun_out/hy_parity.log 2>&1 || { tail -30 gpurun_out/hy_parity.log; echo PARITY_FAILED; exit 1; }
tail -1 gpurun_out/hy_parity.log
for rep in 1 2; do for P in valu hybrid; do
  export STARKHIP_NTT_PATH=$P
  echo "== path $P (round $rep)"
  timeout -k 10 100 python3 tools/ntt_batch_time.py 20 1 8 32 && timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 && timeout -k 10 100 python3 tools/ntt_batch_time.py 21 1 8 && timeout -k 10 100 python3 tools/ntt_batch_time.py 16 64 && timeout -k 10 100 python3 tools/ntt_batch_time.py 19 64 || exit 1
done; done
