cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gaps
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 tools/fri_profile.py 14:1 > gpurun_out/gaps.log 2>&1
f=$(ls gpurun_out/gaps/*/*_kernel_trace.csv | head -1)
python3 tools/trace_gaps.py $f 38 > gpurun_out/gaps_fri14.txt
tail -45 gpurun_out/gaps_fri14.txt
