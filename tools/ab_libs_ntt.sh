# A/B of library builds on the headline transform in ONE session: usage  bash tools/ab_libs_ntt.sh libstarkhip.so libstarkhip_ab1.so ...
# (files under starks_amd/; built with e.g. make -C starks_amd/csrc BUILD=build_ab1 OUT=../libstarkhip_ab1.so EXTRA="-D...")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  export STARKHIP_LIB=$PWD/starks_amd/$L
  timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt_golden or every_size or large_digests or lde or rare_carry or padding" > gpurun_out/ab_parity_$L.log 2>&1 || { tail -20 gpurun_out/ab_parity_$L.log; echo PARITY_FAILED $L; exit 1; }
  echo "parity $L: $(tail -1 gpurun_out/ab_parity_$L.log)"
done
for rep in 1 2 3; do for L in "$@"; do
  export STARKHIP_LIB=$PWD/starks_amd/$L
  timeout -k 10 200 python3 bench.py --no-extras --no-c5 --no-cpu-baseline --no-alu-peak --steps 30 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('== %-22s round $rep  G el/s %.3f  ms/step %.4f  ok %s %s   2^20x1 %.2f  2^20x8 %.2f' % ('$L', d['value']/1e9, d['ms_per_step'], d['check']['roundtrip_ok'], d['check']['matches_fixture'], d['single_vector_elements_per_s']/1e9, d['ntt_2^20_x8_elements_per_s']/1e9))" || exit 1
done; done
for L in "$@"; do
  export STARKHIP_LIB=$PWD/starks_amd/$L
  echo "== per-pass $L"
  rm -rf gpurun_out/abp; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abp -- python3 bench.py --no-extras --no-c5 --no-single --no-cpu-baseline --no-alu-peak --steps 10 --warmup 3 > /dev/null 2>&1
  python3 tools/pass_times.py gpurun_out/abp
done
