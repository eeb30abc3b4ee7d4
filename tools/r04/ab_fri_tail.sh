cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fri or stark or smoke or config5 or two_contexts or busy" > gpurun_out/tail_parity.log 2>&1 || { tail -30 gpurun_out/tail_parity.log; echo PARITY_FAILED; exit 1; }
tail -1 gpurun_out/tail_parity.log
for rep in 1 2; do for T in 13 0 11 15; do
  echo "== STARKHIP_FRI_TAIL_LOG=$T (round $rep)"
  STARKHIP_FRI_TAIL_LOG=$T timeout -k 10 200 python3 tools/fri_profile.py 14:1 16:1 20:1 14:16 16:32 || exit 1
done; done
for T in 13 0; do echo "== c5 STARKHIP_FRI_TAIL_LOG=$T"; STARKHIP_FRI_TAIL_LOG=$T timeout -k 10 200 python3 bench.py --workload c5 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"value": [0-9.]*'; done
