# Evidence session for the headline, ONE box and ONE session (usage: gpurun ... -- 'bash tools/prof_bench.sh r05'): the bench line,
# then the same `bench.py` NTT command under rocprofv3 -- kernel stats, FETCH_SIZE / WRITE_SIZE (separate passes, never with a
# trace domain other than --kernel-trace), SQ counters -- for 2^24 x 1 and 2^20 x 8; the non-NTT workloads' and config 5's kernel
# stats.  Writes gpurun_out/<tag>_*; copy what is to be judged into profiles/<tag>_*.
TAG=${1:-r05}
O=gpurun_out/${TAG}p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf ${O}_*
python3 bench.py > ${O}_bench.json 2> ${O}_bench.err || { tail -5 ${O}_bench.err; exit 1; }
for L in 24 20; do
BT=$([ $L = 24 ] && echo 1 || echo 8)
B="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --no-alu-peak --logn $L --batch $BT --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_stats_$L -- $B > ${O}_stats_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d ${O}_f_$L --pmc FETCH_SIZE -- $B > ${O}_f_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d ${O}_w_$L --pmc WRITE_SIZE -- $B > ${O}_w_$L.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d ${O}_a_$L --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- $B > ${O}_a_$L.log 2>&1 || { echo FAILED $L; tail -3 ${O}_*_$L.log; exit 1; }
python3 tools/make_traffic.py ${O}_f_$L ${O}_w_$L ${O}_stats_$L ${O}_traffic_$L.json $BT $L ${O}_a_$L > /dev/null
python3 tools/pmc_by_grid.py ${O}_a_$L > ${O}_valu_$L.txt
cp $(ls ${O}_stats_$L/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_$L.csv
grep -o '"ms_per_step": [0-9.e+-]*' ${O}_stats_$L.log | head -1
done
tools/alu_mix_bench > ${O}_alu_mix_peak.json
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_non -- python3 tools/nonntt_workload.py 5 > ${O}_non.log 2>&1 || exit 1
cp $(ls ${O}_non/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_nonntt.csv
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_c5 -- python3 bench.py --workload c5 --units 128 --steps 1 --warmup 1 --no-cpu-baseline > ${O}_c5.log 2>&1
cp $(ls ${O}_c5/*/*kernel_stats.csv | head -1) ${O}_kernel_stats_c5.csv
python3 bench.py --workload c5 > ${O}_bench_c5.json 2>> ${O}_bench.err
cat ${O}_alu_mix_peak.json
echo done $?
