cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-extras --no-cpu-baseline"
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
for L in 20 24; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_a_$L --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU -- $B --logn $L --steps 3 --warmup 1 > gpurun_out/pmc_a_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_b_$L --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE -- $B --logn $L --steps 3 --warmup 1 > gpurun_out/pmc_b_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_f_$L --pmc FETCH_SIZE -- $B --logn $L --steps 3 --warmup 1 > gpurun_out/pmc_f_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_w_$L --pmc WRITE_SIZE -- $B --logn $L --steps 3 --warmup 1 > gpurun_out/pmc_w_$L.log 2>&1 || { echo FAILED $L; break; }
done
ls gpurun_out | head -30
