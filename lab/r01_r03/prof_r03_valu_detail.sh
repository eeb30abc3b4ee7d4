# Round 3: where do the waves of the VALU passes wait?  (2^24, one vector)  Two counter passes of 8 SQ counters each.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn 24 --batch 1 --steps 10 --warmup 2"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3o_a --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- $A > gpurun_out/r3o_a.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3o_b --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $A > gpurun_out/r3o_b.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3o_c --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE -- $A > gpurun_out/r3o_c.log 2>&1 || { echo FAILED; exit 1; }
python3 tools/pmc_summary.py gpurun_out/r3o_a gpurun_out/r3o_b gpurun_out/r3o_c | tee gpurun_out/r3o_valu_detail.txt
