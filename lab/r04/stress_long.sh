# Long randomised differential session on the final library (profiles/r04_stress_long.txt): NTT plan variants 8 min, STARK systems 6 min.
# (No pipes into tail: the progress lines must reach gpurun_out/ while the run is going.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "######## stress_plans.py 480 s" > gpurun_out/s2_stress_long.log
timeout -k 10 650 python3 tools/stress_plans.py 480 >> gpurun_out/s2_stress_long.log 2>&1
echo "######## stress_stark.py 360 s" >> gpurun_out/s2_stress_long.log
timeout -k 10 450 python3 tools/stress_stark.py 360 >> gpurun_out/s2_stress_long.log 2>&1
grep -E "^####|^stress:" gpurun_out/s2_stress_long.log
