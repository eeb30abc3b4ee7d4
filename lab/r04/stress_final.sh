# Stress runs on the final library of round 4 (profiles/r04_stress_final.txt): (1) determinism of the 512 proofs of config 5 through two
# contexts, 40 iterations, every byte against the one-context run; (2) random STARK systems against the oracle's coefficient-form prover;
# (3) random NTT plan / tile / tile-order variants against oracle/oracle.c; (4) the two-context and busy-second-stream tests ten times.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "######## c5_repro: 40 iterations x 512 proofs through two contexts"
timeout -k 10 400 python3 tools/r04/c5_repro.py --iters 40 --quiet | tail -4
echo "######## c5_repro: 10 iterations, 3 contexts, chunks of 64"
timeout -k 10 300 python3 tools/r04/c5_repro.py --iters 10 --quiet --streams 3 --chunk 64 | tail -3
echo "######## stress_stark.py 240 s"
timeout -k 10 400 python3 tools/stress_stark.py 240 | tail -3
echo "######## stress_plans.py 180 s"
timeout -k 10 400 python3 tools/stress_plans.py 180 | tail -3
echo "######## two-context / busy-stream tests x 10"
for i in 1 2 3 4 5 6 7 8 9 10; do timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -q -x -k "two_contexts_running or busy_second" -p no:cacheprovider 2>&1 | tail -1; done
