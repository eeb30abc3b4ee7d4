# Long randomised differential session on the final library (profiles/r04_stress_long.txt): NTT plan variants 9 min, STARK systems 7 min.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "######## stress_plans.py 540 s"; timeout -k 10 700 python3 tools/stress_plans.py 540 | tail -2
echo "######## stress_stark.py 420 s"; timeout -k 10 500 python3 tools/stress_stark.py 420 | tail -2
