set -e
SEL="test_ntt_golden_vectors or test_ntt_every_size_vs_oracle or test_ntt_padding_and_batch or test_randomized_ntt_differential or test_ntt_large_digests_vs_oracle_fixture or test_lde_golden or test_fri_proofs_golden or test_stark_proofs_golden or test_device_resident_pipeline or test_rare_carry_branches"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "$SEL" -p no:cacheprovider > gpurun_out/plan_par.txt 2>&1 || { tail -30 gpurun_out/plan_par.txt; echo "PARITY FAILED"; exit 1; }
tail -1 gpurun_out/plan_par.txt
timeout -k 10 120 python tools/ntt_batch_time.py 20 1 8 32
timeout -k 10 120 python tools/ntt_batch_time.py 19 64
timeout -k 10 120 python tools/ntt_batch_time.py 24 1
timeout -k 10 120 python tools/ntt_batch_time.py 22 4
timeout -k 10 120 python tools/ntt_batch_time.py 16 64
python3 bench.py --workload c5 --no-cpu-baseline | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', l['value'])"
