# Round-3 evidence, all from ONE box and ONE session: the bench line, then the same `bench.py` command under rocprofv3 --
# kernel stats, FETCH_SIZE / WRITE_SIZE (separate passes), VALU counters -- for the headline (one 2^24-point vector) and for
# 8 x 2^20; the FRI commit of a 2^20-step trace, the 2^24-leaf Merkle commit and config 5 under the kernel trace.
# (send this script's own output to a file that does NOT match gpurun_out/r3p_*: the next line removes those)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r3p_*
python3 bench.py > gpurun_out/r3p_bench.json 2> gpurun_out/r3p_bench.err || { tail -5 gpurun_out/r3p_bench.err; exit 1; }
for L in 24 20; do
BT=$([ $L = 24 ] && echo 1 || echo 8)
B="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn $L --batch $BT --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3p_stats_$L -- $B > gpurun_out/r3p_stats_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3p_f_$L --pmc FETCH_SIZE -- $B > gpurun_out/r3p_f_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3p_w_$L --pmc WRITE_SIZE -- $B > gpurun_out/r3p_w_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3p_a_$L --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- $B > gpurun_out/r3p_a_$L.log 2>&1 || { echo FAILED $L; exit 1; }
python3 tools/make_traffic.py gpurun_out/r3p_f_$L gpurun_out/r3p_w_$L gpurun_out/r3p_stats_$L gpurun_out/r3p_traffic_$L.json $BT $L > /dev/null
python3 tools/pmc_summary.py gpurun_out/r3p_a_$L > gpurun_out/r3p_valu_$L.txt
cp $(ls gpurun_out/r3p_stats_$L/*/*kernel_stats.csv | head -1) gpurun_out/r3p_kernel_stats_$L.csv
grep -o '"ms_per_step": [0-9.e+-]*' gpurun_out/r3p_stats_$L.log | head -1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3p_fri20 -- python3 tools/fri_profile.py 20:1 > gpurun_out/r3p_fri20.log 2>&1 || exit 1
cp $(ls gpurun_out/r3p_fri20/*/*kernel_stats.csv | head -1) gpurun_out/r3p_kernel_stats_fri20.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3p_mk24 -- python3 tools/merkle_time.py > gpurun_out/r3p_mk24.log 2>&1 || exit 1
cp $(ls gpurun_out/r3p_mk24/*/*kernel_stats.csv | head -1) gpurun_out/r3p_kernel_stats_mk24.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3p_c5 -- python3 bench.py --workload c5 --units 128 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3p_c5.log 2>&1
cp $(ls gpurun_out/r3p_c5/*/*kernel_stats.csv | head -1) gpurun_out/r3p_kernel_stats_c5.csv
python3 bench.py --workload c5 > gpurun_out/r3p_bench_c5.json 2>> gpurun_out/r3p_bench.err
echo done $?
