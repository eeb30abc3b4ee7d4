# Profiles for profiles/: kernel stats + PMC (separate passes) of the default bench workload (2^20 fwd+inv NTT).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_stats -- $B > gpurun_out/fin_stats.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_f --pmc FETCH_SIZE -- $B > gpurun_out/fin_f.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_w --pmc WRITE_SIZE -- $B > gpurun_out/fin_w.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_a --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU -- $B > gpurun_out/fin_a.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_b --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE -- $B > gpurun_out/fin_b.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_full -- python3 bench.py --no-cpu-baseline --steps 10 > gpurun_out/fin_full.log 2>&1
echo done $?
