# Round 3: per-pass kernel durations of the 2^24 transform, both paths, with and without the 2^24-entry row table of pass 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in valu mfma; do for TW in 23 24; do
export STARKHIP_NTT_PATH=$P STARKHIP_TW2_MAX_LOG=$TW
A="python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --logn 24 --batch 1 --steps 10 --warmup 2"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3g_${P}_$TW -- $A > gpurun_out/r3g_${P}_$TW.log 2>&1 || { echo FAILED; tail -5 gpurun_out/r3g_${P}_$TW.log; exit 1; }
echo "== path=$P tw2_max_log=$TW"; grep -o '"value": [0-9.e+]*' gpurun_out/r3g_${P}_$TW.log | head -1
python3 tools/pass_times.py gpurun_out/r3g_${P}_$TW
done; done
