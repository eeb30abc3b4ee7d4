# Stress session on the library as it stands (usage: gpurun ... -- 'bash tools/stress_session.sh > gpurun_out/stress.txt 2>&1'):
# (1) determinism of config 5's 512 proofs through two and three contexts, every byte against the one-context run; (2) random STARK systems
# against the oracle's coefficient-form prover; (3) random NTT plan / tile variants against oracle/oracle.c; (4) the two-context and
# busy-second-stream tests five times.  Steps are joined with &&: a failure stops the session.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "######## c5_repro: 30 iterations x 512 proofs through two contexts" &&
timeout -k 10 500 python3 tools/c5_repro.py --iters 30 --quiet | tail -4 &&
echo "######## c5_repro: 10 iterations, 3 contexts, chunks of 64" &&
timeout -k 10 300 python3 tools/c5_repro.py --iters 10 --quiet --streams 3 --chunk 64 | tail -3 &&
echo "######## stress_stark.py 180 s" &&
timeout -k 10 400 python3 tools/stress_stark.py 180 | tail -3 &&
echo "######## stress_plans.py 120 s" &&
timeout -k 10 400 python3 tools/stress_plans.py 120 | tail -3 &&
echo "######## two-context / busy-stream tests x 5" &&
for i in 1 2 3 4 5; do timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -q -x -k "two_contexts_running or busy_second" -p no:cacheprovider 2>&1 | tail -1 || exit 1; done
echo "######## done"
