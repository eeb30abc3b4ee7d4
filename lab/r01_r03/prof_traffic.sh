cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3"
rm -rf gpurun_out/fin_stats gpurun_out/fin_f gpurun_out/fin_w
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_stats -- $B > gpurun_out/fin_stats.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_f --pmc FETCH_SIZE -- $B > gpurun_out/fin_f.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fin_w --pmc WRITE_SIZE -- $B > gpurun_out/fin_w.log 2>&1
echo done $?
python3 bench.py --no-extras --no-cpu-baseline | cut -c1-200
