# Round 3: phase timeline (s_memtime stamps, diagnostic library) and issue / wait counters of the MFMA tile pass with the
# generated asm stages.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_stamps.so STARKHIP_NTT_PATH=mfma
  for P in 0 1 2; do echo "== 2^20 x 8 vectors, pass $P"; STARKHIP_STAMP_PASS=$P timeout -k 10 120 python3 tools/mfma_phases.py 20 8 || exit 1; done
  for P in 0 1 2; do echo "== 2^24, pass $P"; STARKHIP_STAMP_PASS=$P timeout -k 10 120 python3 tools/mfma_phases.py 24 1 || exit 1; done ) > gpurun_out/r3c_phases.txt 2>&1
cat gpurun_out/r3c_phases.txt
export STARKHIP_NTT_PATH=mfma
for L in 20 24; do
BT=$([ $L = 24 ] && echo 1 || echo 8)
A="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn $L --batch $BT --steps 10 --warmup 2"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3c_a_$L --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $A > gpurun_out/r3c_a_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3c_b_$L --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE -- $A > gpurun_out/r3c_b_$L.log 2>&1 || { echo PMC_FAILED $L; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/r3c_a_20 gpurun_out/r3c_b_20 gpurun_out/r3c_a_24 gpurun_out/r3c_b_24 > gpurun_out/r3c_pmc.txt 2>&1
cat gpurun_out/r3c_pmc.txt
