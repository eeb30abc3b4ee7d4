# PMC view of the MFMA tile passes at 2^24 (one vector) and 2^20 (8 vectors): issue / wait split, MFMA busy, traffic.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-extras --no-cpu-baseline --no-c5"
for L in 24 20; do
BT=$([ $L = 24 ] && echo 1 || echo 8)
A="$B --logn $L --batch $BT --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mf_s_$L -- $A > gpurun_out/mf_s_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mf_a_$L --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $A > gpurun_out/mf_a_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mf_b_$L --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE -- $A > gpurun_out/mf_b_$L.log 2>&1 || { echo FAILED $L; break; }
done
python3 tools/pmc_summary.py gpurun_out/mf_a_24 gpurun_out/mf_b_24 gpurun_out/mf_a_20 gpurun_out/mf_b_20 > gpurun_out/mf_summary.txt 2>&1
cat gpurun_out/mf_summary.txt
