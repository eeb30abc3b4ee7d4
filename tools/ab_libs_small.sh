# A/B of library builds on the SMALL, latency-bound shapes (2^14 / 2^16-step FRI commits, single STARK proofs, small transforms) in ONE
# session: usage  bash tools/ab_libs_small.sh libstarkhip.so libstarkhip_ab0.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  export STARKHIP_LIB=$PWD/starks_amd/$L
  timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_reference_suite.py -m gpu -x -q -k "ntt or fri or stark or lde or reference" > gpurun_out/ab_parity_$L.log 2>&1 || { tail -20 gpurun_out/ab_parity_$L.log; echo PARITY_FAILED $L; exit 1; }
  echo "parity $L: $(tail -1 gpurun_out/ab_parity_$L.log)"
done
for rep in 1 2 3; do for L in "$@"; do
  export STARKHIP_LIB=$PWD/starks_amd/$L
  echo "== $L round $rep"
  timeout -k 10 200 python3 tools/fri_profile.py 10:1 12:1 14:1 14:4 16:1 | grep steps | sed 's/^/fri   /' || exit 1
  timeout -k 10 200 python3 tools/stark_time.py 10:1 12:1 14:1 16:1 | grep steps | sed 's/^/stark /' || exit 1
  timeout -k 10 200 python3 tools/ntt_batch_time.py 14 1 2 | grep -i "batch" | sed 's/^/ntt   /'
  timeout -k 10 200 python3 tools/ntt_batch_time.py 16 1 | grep -i "batch" | sed 's/^/ntt   /'
done; done
