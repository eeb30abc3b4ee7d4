# A/B in one session: the tree's library (A) against starks_amd/libstarkhip_ab.so (B) on the Merkle / FRI / config-5 timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "merkle or fri or stark" > gpurun_out/abm_parity.log 2>&1 || { tail -20 gpurun_out/abm_parity.log; echo PARITY_FAILED; exit 1; }
tail -1 gpurun_out/abm_parity.log
for rep in 1 2; do for L in A B; do
  if [ $L = B ]; then export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_ab.so; else unset STARKHIP_LIB; fi
  echo "== lib $L (round $rep)"
  timeout -k 10 100 python3 tools/merkle_time.py && timeout -k 10 200 python3 tools/fri_profile.py 20:1 14:1 && timeout -k 10 200 python3 bench.py --workload c5 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"value": [0-9.]*' || exit 1
done; done
