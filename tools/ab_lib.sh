# A/B of two builds of the library in one session: starks_amd/libstarkhip.so (A, the tree) vs starks_amd/libstarkhip_ab.so (B)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or lde or fri_proofs_golden or stark_proofs_golden or rare_carry or random_plan" > gpurun_out/ab_parity.log 2>&1 || { tail -20 gpurun_out/ab_parity.log; echo PARITY_FAILED; exit 1; }
tail -1 gpurun_out/ab_parity.log
for rep in 1 2; do for L in A B; do
  if [ $L = B ]; then export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_ab.so; else unset STARKHIP_LIB; fi
  echo "== lib $L (round $rep)"
  timeout -k 10 100 python3 tools/ntt_batch_time.py 20 1 8 32 && timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 && timeout -k 10 100 python3 tools/ntt_batch_time.py 19 64 || exit 1
done; done
