# non-temporal loads in pass 0 of the 2^24 transform.  A = default, B1 = nt on the 512 MiB tw2 row-table stream,
# B2 = nt on the pass's own element loads.  One session: parity (the 2^24 digest) per build, then alternating timings, per-pass
# durations from a kernel trace, and FETCH_SIZE / WRITE_SIZE of passes 0 and 1 per build.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
lib() { case $1 in A) unset STARKHIP_LIB;; B1) export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_ab.so;; B2) export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_ab2.so;; esac; }
for L in B1 B2; do lib $L; timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_digests and 24" 2>&1 | tail -1; done
for rep in 1 2 3; do for L in A B1 B2; do lib $L; echo "== $L (round $rep)"; timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 2 || exit 1; done; done
B="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn 24 --batch 1 --steps 10 --warmup 2"
for L in A B1 B2; do lib $L; echo "== $L per-pass durations (kernel trace) and HBM-side bytes"
  rm -rf gpurun_out/nt_$L*; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/nt_${L}_t -- $B > /dev/null 2>&1 && python3 tools/pass_times.py gpurun_out/nt_${L}_t
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/nt_${L}_f --pmc FETCH_SIZE -- $B > /dev/null 2>&1 &&
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/nt_${L}_w --pmc WRITE_SIZE -- $B > /dev/null 2>&1 &&
  python3 tools/pmc_by_grid.py gpurun_out/nt_${L}_f gpurun_out/nt_${L}_w | grep "^ntt_pass\|HBM-side"
done
