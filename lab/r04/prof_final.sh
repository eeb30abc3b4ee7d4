# Round-4 final evidence, ONE box, ONE session (after tools/r04/prof_bench.sh has produced the traffic / VALU JSONs that bench.py reads):
# the default bench line and the c5 line, the non-NTT counters with the final kernels, the c5 kernel stats, the FRI timelines.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err || { tail -5 gpurun_out/r4f_bench.err; exit 1; }
python3 bench.py --workload c5 > gpurun_out/r4f_bench_c5.json 2>> gpurun_out/r4f_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f_c5 -- python3 bench.py --workload c5 --units 128 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r4f_c5.log 2>&1
cp $(ls gpurun_out/r4f_c5/*/*kernel_stats.csv | head -1) gpurun_out/r4f_c5_128proofs_kernel_stats.csv
bash tools/r04/prof_nonntt_pmc.sh > gpurun_out/r4f_nonntt.log 2>&1 || { tail -5 gpurun_out/r4f_nonntt.log; exit 1; }
python3 tools/pmc_by_grid.py gpurun_out/r04n/a gpurun_out/r04n/b gpurun_out/r04n/c gpurun_out/r04n/d gpurun_out/r04n/e > gpurun_out/r4f_nonntt_pmc_by_kernel.txt
bash tools/r04/fri_trace.sh 14 > gpurun_out/r4f_fri_trace_14.txt 2>&1
bash tools/r04/fri_trace.sh 20 > gpurun_out/r4f_fri_trace_20.txt 2>&1
bash tools/r04/merkle_trace.sh > gpurun_out/r4f_merkle_trace.txt 2>&1
python3 - <<'P'
import json
b = json.load(open("gpurun_out/r4f_bench.json"))
print("value %.4g el/s, %.4f ms/step, roofline.frac %.4f, roofline_alu %s" % (b["value"], b["ms_per_step"], b["roofline"]["frac"], {k: b.get("roofline_alu", {}).get(k) for k in ("achieved", "peak", "frac")}))
print({k: b[k] for k in ("single_vector_elements_per_s", "ntt_2^20_x8_elements_per_s", "c5_proofs_per_s", "fri_commit_ms_2^20_trace", "fri_commit_ms_2^14_trace")})
print({k: (v.get("ms") or v.get("ms_per_batch")) for k, v in b["extra"].items()})
c = json.load(open("gpurun_out/r4f_bench_c5.json"))
print("c5 %.1f proofs/s" % c["value"])
P
