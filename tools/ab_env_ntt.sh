# A/B of environment knobs on the headline transform in ONE session: usage  bash tools/ab_env_ntt.sh "A=" "STARKHIP_X=1" ...
# each argument is an env assignment list (use "base=" for the default); bench.py --no-extras/--no-c5 NTT value + per-pass times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for V in "$@"; do
  echo "== [$V] (round $rep)"
  env $V timeout -k 10 200 python3 bench.py --no-extras --no-c5 --no-single --no-cpu-baseline --no-alu-peak --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   G el/s %.3f  ms/step %.4f  ok %s %s' % (d['value']/1e9, d['ms_per_step'], d['check']['roundtrip_ok'], d['check']['matches_fixture']))" || exit 1
done; done
for V in "$@"; do
  echo "== per-pass [$V]"
  rm -rf gpurun_out/abp; env $V rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abp -- python3 bench.py --no-extras --no-c5 --no-single --no-cpu-baseline --no-alu-peak --steps 10 --warmup 3 > /dev/null 2>&1
  python3 tools/pass_times.py gpurun_out/abp
done
