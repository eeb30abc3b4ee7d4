# Round 3: instruction-cache counters of the tile passes (is the 90 KB straight-line MFMA kernel fetch-bound?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in mfma valu; do
export STARKHIP_NTT_PATH=$P
A="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn 20 --batch 8 --steps 10 --warmup 2"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3e_ic_$P --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -- $A > gpurun_out/r3e_ic_$P.log 2>&1 || { echo FAILED $P; tail -5 gpurun_out/r3e_ic_$P.log; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/r3e_ic_mfma gpurun_out/r3e_ic_valu > gpurun_out/r3e_icache.txt 2>&1
cat gpurun_out/r3e_icache.txt
