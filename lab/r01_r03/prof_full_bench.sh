cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fin_full
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin_full -- python3 bench.py --steps 50 --warmup 5 > gpurun_out/fin_full.log 2>&1
echo $?
tail -1 gpurun_out/fin_full.log | cut -c1-300
