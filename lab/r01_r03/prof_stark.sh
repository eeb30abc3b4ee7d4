cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/stark_time.py 14:1 16:1 16:16 16:32 20:1 > gpurun_out/stark_time.log 2>&1 || { tail -20 gpurun_out/stark_time.log; exit 1; }
cat gpurun_out/stark_time.log
rm -rf gpurun_out/stark_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stark_prof -- python3 tools/stark_time.py 16:16 > gpurun_out/stark_prof.log 2>&1
echo rc=$?
