# End-of-round evidence: bench JSON, kernel stats (2^20 workload and full extras), PMC traffic + VALU counters.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3"
rm -rf gpurun_out/fin_*
python3 bench.py > gpurun_out/fin_bench.json 2> gpurun_out/fin_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_stats -- $B > gpurun_out/fin_stats.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_f --pmc FETCH_SIZE -- $B > gpurun_out/fin_f.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_w --pmc WRITE_SIZE -- $B > gpurun_out/fin_w.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_a --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $B > gpurun_out/fin_a.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_b --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM -- $B > gpurun_out/fin_b.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_fri -- python3 tools/fri_profile.py 14:1 20:1 16:32 > gpurun_out/fin_fri.log 2>&1
echo done $?
