cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_c_24
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_c_24 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --no-extras --no-cpu-baseline --logn 24 --steps 3 --warmup 1 > gpurun_out/pmc_c_24.log 2>&1
echo $?
