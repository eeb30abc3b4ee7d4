# Plans of transforms above 2^24 points in ONE session: usage  bash tools/ab_plans_big.sh 25 26 27 28
# per size: the default plan, the three-pass plan of radices <= 2^10, each with the row-major inter-pass tables (tw2) up to the full size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # logn, env assignments
  L=$1; shift
  env "$@" timeout -k 10 300 python3 bench.py --logn $L --no-extras --no-c5 --no-single --no-cpu-baseline --no-alu-peak --steps 8 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   G el/s %.3f  ms/step %.3f  ok %s' % (d['value']/1e9, d['ms_per_step'], d['check']['roundtrip_ok']))" || exit 1
}
for L in "$@"; do
  a=$(( (L + 2) / 3 )); b=$(( (L - a + 1) / 2 )); c=$(( L - a - b ))
  for V in "X=0" "STARKHIP_TW2_MAX_LOG=$L" "STARKHIP_NTT_RADICES=$a,$b,$c" "STARKHIP_NTT_RADICES=$a,$b,$c STARKHIP_TW2_MAX_LOG=$L"; do
    echo "== 2^$L [$V]"
    run $L $V
  done
done
