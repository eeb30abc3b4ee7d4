set -e
SEL="test_ntt_golden_vectors or test_ntt_every_size_vs_oracle or test_ntt_padding_and_batch or test_ntt_large_digests_vs_oracle_fixture or test_lde_golden or test_fri_proofs_golden or test_stark_proofs_golden or test_rare_carry_branches or test_fold_golden or test_randomized_ntt_differential"
STARKHIP_LIB=$PWD/starks_amd/libstarkhip_B.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "$SEL" -p no:cacheprovider > gpurun_out/plan_par.txt 2>&1 || { tail -30 gpurun_out/plan_par.txt; echo "PARITY FAILED"; exit 1; }
tail -1 gpurun_out/plan_par.txt
for rep in 1 2; do
for lib in A B C; do
echo "== lib $lib"
STARKHIP_LIB=$PWD/starks_amd/libstarkhip_$lib.so timeout -k 10 120 python tools/ntt_batch_time.py 20 1 8 32
STARKHIP_LIB=$PWD/starks_amd/libstarkhip_$lib.so timeout -k 10 120 python tools/ntt_batch_time.py 19 64
STARKHIP_LIB=$PWD/starks_amd/libstarkhip_$lib.so timeout -k 10 120 python tools/ntt_batch_time.py 24 1
done
done
