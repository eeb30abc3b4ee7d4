# Round 4: counter-level evidence for the non-NTT kernels: kernel stats + separate --pmc passes of the
# same command.  Output: gpurun_out/r04_nonntt_*.  (--pmc only with --kernel-trace; one rocprofv3 run per counter set.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="python3 tools/r04/nonntt_workload.py 3"
O=gpurun_out/r04n
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $A > $O/kt.log 2>&1 || { cat $O/kt.log; exit 1; }
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) gpurun_out/r04_nonntt_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $O/a --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- $A > $O/a.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d $O/b --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $A > $O/b.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d $O/c --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -- $A > $O/c.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d $O/d --pmc FETCH_SIZE -- $A > $O/d.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d $O/e --pmc WRITE_SIZE -- $A > $O/e.log 2>&1 || { echo FAILED; tail -5 $O/*.log; exit 1; }
python3 tools/pmc_summary.py $O/a $O/b $O/c $O/d $O/e > gpurun_out/r04_nonntt_pmc.txt
cat $O/kt.log
tail -120 gpurun_out/r04_nonntt_pmc.txt
