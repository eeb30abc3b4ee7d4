# A/B of library builds on single STARK proofs (latency) in ONE session: usage  bash tools/ab_libs_stark.sh libstarkhip.so libstarkhip_ab0.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  export STARKHIP_LIB=$PWD/starks_amd/$L
  timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_suite.py -m gpu -x -q -k "stark" > gpurun_out/ab_parity_$L.log 2>&1 || { tail -20 gpurun_out/ab_parity_$L.log; echo PARITY_FAILED $L; exit 1; }
  echo "parity $L: $(tail -1 gpurun_out/ab_parity_$L.log)"
done
for rep in 1 2 3; do for L in "$@"; do
  export STARKHIP_LIB=$PWD/starks_amd/$L
  echo "== $L round $rep"
  timeout -k 10 200 python3 tools/stark_time.py 8:1 10:1 12:1 14:1 16:1 17:1 12:8 14:4 14:8 16:2 | grep steps || exit 1
done; done
