# Round-2 evidence, all from ONE box: bench JSON, then the same `bench.py` command under rocprofv3 -- kernel stats,
# FETCH_SIZE / WRITE_SIZE (separate passes), VALU counters -- for the 2^20 x 8 workload and for one 2^24 vector.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r2f_*
python3 bench.py > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err
for L in 20 24; do
BT=$([ $L = 24 ] && echo 1 || echo 8)
B="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn $L --batch $BT --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f_stats_$L -- $B > gpurun_out/r2f_stats_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2f_f_$L --pmc FETCH_SIZE -- $B > gpurun_out/r2f_f_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2f_w_$L --pmc WRITE_SIZE -- $B > gpurun_out/r2f_w_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2f_a_$L --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- $B > gpurun_out/r2f_a_$L.log 2>&1 || { echo FAILED $L; exit 1; }
python3 tools/make_traffic.py gpurun_out/r2f_f_$L gpurun_out/r2f_w_$L gpurun_out/r2f_stats_$L gpurun_out/r2f_traffic_$L.json $BT $L > /dev/null
python3 tools/pmc_summary.py gpurun_out/r2f_a_$L > gpurun_out/r2f_valu_$L.txt
cp $(ls gpurun_out/r2f_stats_$L/*/*kernel_stats.csv | head -1) gpurun_out/r2f_kernel_stats_$L.csv
done
# the many-proof workload (config 5) under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2f_c5 -- python3 bench.py --workload c5 --units 128 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r2f_c5.log 2>&1
cp $(ls gpurun_out/r2f_c5/*/*kernel_stats.csv | head -1) gpurun_out/r2f_kernel_stats_c5.csv
python3 bench.py --workload c5 > gpurun_out/r2f_bench_c5.json 2>> gpurun_out/r2f_bench.err
# the two NTT paths side by side, and the MFMA tile pass's phase timeline + issue / wait split
( for P in valu mfma; do echo "== STARKHIP_NTT_PATH=$P"; STARKHIP_NTT_PATH=$P python3 tools/ntt_batch_time.py 20 1 8 32; STARKHIP_NTT_PATH=$P python3 tools/ntt_batch_time.py 24 1; STARKHIP_NTT_PATH=$P python3 tools/ntt_batch_time.py 16 64; STARKHIP_NTT_PATH=$P python3 tools/ntt_batch_time.py 19 64; done ) > gpurun_out/r2f_ntt_paths.txt 2>&1
( export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_stamps.so STARKHIP_NTT_PATH=mfma; for P in 0 1 2; do echo "== 2^20 x 8 vectors, pass $P"; STARKHIP_STAMP_PASS=$P python3 tools/mfma_phases.py 20 8; done; for P in 0 2; do echo "== 2^24, pass $P"; STARKHIP_STAMP_PASS=$P python3 tools/mfma_phases.py 24 1; done ) > gpurun_out/r2f_mfma_phases.txt 2>&1
export STARKHIP_NTT_PATH=mfma
A="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn 20 --batch 8 --steps 20 --warmup 3"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2f_mfma_a --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $A > gpurun_out/r2f_mfma_a.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2f_mfma_b --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU GRBM_GUI_ACTIVE -- $A > gpurun_out/r2f_mfma_b.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r2f_mfma_a gpurun_out/r2f_mfma_b > gpurun_out/r2f_mfma_pmc.txt 2>&1
unset STARKHIP_NTT_PATH
python3 tools/pcie_rate.py > gpurun_out/r2f_pcie.txt 2>&1
echo done $?
