# the LDS-resident matrix-core tile: real HBM traffic per pass (rocprofv3 counters) and what the row table of the inter-pass
# twiddles costs (STARKHIP_TW2_MAX_LOG=0: two small tables and a second product instead of one 32-byte load per element)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export STARKHIP_NTT_PATH=mfma_lds
for SH in "20 8" "24 1"; do
  A="python3 tools/ntt_batch_time.py $SH"
  rm -rf gpurun_out/lt_s gpurun_out/lt_f gpurun_out/lt_w
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lt_s -- $A > gpurun_out/lt_s.log 2>&1 &&
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lt_f --pmc FETCH_SIZE -- $A > gpurun_out/lt_f.log 2>&1 &&
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lt_w --pmc WRITE_SIZE -- $A > gpurun_out/lt_w.log 2>&1 || { tail -5 gpurun_out/lt_s.log; exit 1; }
  echo "== mfma_lds, 2^$SH"; python3 tools/kernel_traffic.py gpurun_out/lt_s gpurun_out/lt_f gpurun_out/lt_w
done
rm -rf gpurun_out/lt_s gpurun_out/lt_f gpurun_out/lt_w
for TW in 23 0; do for SW in 1 3; do
  echo "== STARKHIP_TW2_MAX_LOG=$TW STARKHIP_XCD_SWZ=$SW"
  STARKHIP_TW2_MAX_LOG=$TW STARKHIP_XCD_SWZ=$SW timeout -k 10 100 python3 tools/ntt_batch_time.py 20 1 8 32 && STARKHIP_TW2_MAX_LOG=$TW STARKHIP_XCD_SWZ=$SW timeout -k 10 100 python3 tools/ntt_batch_time.py 24 1 || exit 1
done; done
