# where the waves of the hybrid tile pass spend their cycles, the VALU passes beside it (2^20 x 8): two counter passes each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="python3 tools/ntt_batch_time.py 20 8"
for P in valu hybrid; do
export STARKHIP_NTT_PATH=$P
rm -rf gpurun_out/hp_a gpurun_out/hp_b
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hp_a --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- $A > gpurun_out/hp_a.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/hp_b --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $A > gpurun_out/hp_b.log 2>&1 || { echo FAILED; tail -5 gpurun_out/hp_b.log; exit 1; }
echo "#### path $P"; python3 tools/pmc_summary.py gpurun_out/hp_a gpurun_out/hp_b
done
rm -rf gpurun_out/hp_a gpurun_out/hp_b
