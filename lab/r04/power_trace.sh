# Samples rocm-smi (package power, sclk, junction temperature) twice a second while a SUSTAINED workload runs for several seconds:
# (1) 3000 NTT pairs of 2^24 points (about 7.5 s), (2) 60 steps of config 5 (about 6 s).  profiles/r04_power_and_clocks.txt.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sample() { for i in $(seq 1 $1); do echo "t=$(date +%s.%N | cut -c7-14) $(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|Sensor junction" | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.4; done; }
echo "== idle"; sample 2
echo "== bench.py --logn 24 --steps 3000 (NTT pairs)"
python3 bench.py --no-extras --no-cpu-baseline --no-c5 --no-single --steps 3000 --warmup 3 > gpurun_out/pw_ntt.json 2>/dev/null &
P=$!; sample 36; wait $P; grep -o '"value": [0-9.e+]*' gpurun_out/pw_ntt.json; grep -o '"ms_per_step": [0-9.e+]*' gpurun_out/pw_ntt.json
echo "== config 5, 60 steps"
python3 bench.py --workload c5 --steps 60 --no-cpu-baseline --no-extras > gpurun_out/pw_c5.json 2>/dev/null &
P=$!; sample 40; wait $P; grep -o '"value": [0-9.e+]*' gpurun_out/pw_c5.json
