# VALU issue / wait counters of the bench.py NTT command (2^20 x 8) for the default plan and for STARKHIP_NTT_RADICES=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn 20 --batch 8 --steps 20 --warmup 3"
rm -rf gpurun_out/pv_a gpurun_out/pv_b
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pv_a --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- $B > gpurun_out/pv_a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pv_b --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM -- $B > gpurun_out/pv_b.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/pv_a gpurun_out/pv_b
