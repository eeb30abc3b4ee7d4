# L2 request volume of the MFMA tile pass (operand images come from L2 for every butterfly)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "TC[CP]_[A-Z_0-9]*" | sort -u | tr "\n" " " > gpurun_out/r3r_tc_counters.txt
for P in mfma valu; do
export STARKHIP_NTT_PATH=$P
A="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn 20 --batch 8 --steps 10 --warmup 2"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3r_l2_$P --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum -- $A > gpurun_out/r3r_l2_$P.log 2>&1 || { echo FAILED $P; tail -5 gpurun_out/r3r_l2_$P.log; }
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3r_l1_$P --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum -- $A > gpurun_out/r3r_l1_$P.log 2>&1 || { echo FAILED l1 $P; tail -5 gpurun_out/r3r_l1_$P.log; }
done
python3 tools/pmc_summary.py gpurun_out/r3r_l2_mfma gpurun_out/r3r_l1_mfma gpurun_out/r3r_l2_valu gpurun_out/r3r_l1_valu 2>&1 | tee gpurun_out/r3r_l2.txt
