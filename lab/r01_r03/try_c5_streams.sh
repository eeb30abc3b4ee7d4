# config 5 with the batched launches dealt to 1 / 2 / 3 library contexts (streams), chunk sizes 128 and 64
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for S in "1 128" "2 128" "2 64" "3 64" "4 32"; do set -- $S
  echo "== streams $1, $2 proofs per launch (round $rep)"
  timeout -k 10 300 python3 bench.py --workload c5 --no-cpu-baseline --c5-streams $1 --chunk $2 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['c5']['check']['ok'])" || exit 1
done; done
