# kernel trace of one FRI commit of a 2^20-step trace and of the 2^24-leaf Merkle commit
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fri20 gpurun_out/mk24
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fri20 -- python3 tools/fri_profile.py 20:1 > gpurun_out/fri20.log 2>&1 || exit 1
cp $(ls gpurun_out/fri20/*/*kernel_stats.csv | head -1) gpurun_out/fri20_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mk24 -- python3 tools/merkle_time.py > gpurun_out/mk24.log 2>&1 || exit 1
cp $(ls gpurun_out/mk24/*/*kernel_stats.csv | head -1) gpurun_out/mk24_kernel_stats.csv
cat gpurun_out/fri20.log gpurun_out/mk24.log
