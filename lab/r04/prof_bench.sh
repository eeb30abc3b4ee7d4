# Round-4 evidence for the headline, ONE box and ONE session: the bench line, then the same `bench.py` NTT command under rocprofv3 --
# kernel stats, FETCH_SIZE / WRITE_SIZE (separate passes), SQ_INSTS_VALU -- for 2^24 x 1 and 2^20 x 8; then the non-NTT workloads'
# kernel stats.  Writes gpurun_out/r4p_*; copy what is to be judged into profiles/r04_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4p_*
python3 bench.py > gpurun_out/r4p_bench.json 2> gpurun_out/r4p_bench.err || { tail -5 gpurun_out/r4p_bench.err; exit 1; }
for L in 24 20; do
BT=$([ $L = 24 ] && echo 1 || echo 8)
B="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn $L --batch $BT --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4p_stats_$L -- $B > gpurun_out/r4p_stats_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4p_f_$L --pmc FETCH_SIZE -- $B > gpurun_out/r4p_f_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4p_w_$L --pmc WRITE_SIZE -- $B > gpurun_out/r4p_w_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4p_a_$L --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- $B > gpurun_out/r4p_a_$L.log 2>&1 || { echo FAILED $L; tail -3 gpurun_out/r4p_*_$L.log; exit 1; }
python3 tools/make_traffic.py gpurun_out/r4p_f_$L gpurun_out/r4p_w_$L gpurun_out/r4p_stats_$L gpurun_out/r4p_traffic_$L.json $BT $L gpurun_out/r4p_a_$L > /dev/null
python3 tools/pmc_by_grid.py gpurun_out/r4p_a_$L > gpurun_out/r4p_valu_$L.txt
cp $(ls gpurun_out/r4p_stats_$L/*/*kernel_stats.csv | head -1) gpurun_out/r4p_kernel_stats_$L.csv
grep -o '"ms_per_step": [0-9.e+-]*' gpurun_out/r4p_stats_$L.log | head -1
done
tools/alu_mix_bench > gpurun_out/r4p_alu_mix_peak.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4p_non -- python3 tools/r04/nonntt_workload.py 5 > gpurun_out/r4p_non.log 2>&1 || exit 1
cp $(ls gpurun_out/r4p_non/*/*kernel_stats.csv | head -1) gpurun_out/r4p_kernel_stats_nonntt.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4p_c5 -- python3 bench.py --workload c5 --units 128 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r4p_c5.log 2>&1
cp $(ls gpurun_out/r4p_c5/*/*kernel_stats.csv | head -1) gpurun_out/r4p_kernel_stats_c5.csv
python3 bench.py --workload c5 > gpurun_out/r4p_bench_c5.json 2>> gpurun_out/r4p_bench.err
cat gpurun_out/r4p_alu_mix_peak.json gpurun_out/r4p_non.log | grep -v "^[WE]2026"
echo done $?
