# Round 3: the three tile-pass paths side by side on one box (VALU passes, MFMA passes with the C++ butterflies, MFMA passes
# with the generated asm stages), after the parity suite of the MFMA path.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_mfma_tile_passes_parity" > gpurun_out/r3b_parity.log 2>&1 || { tail -30 gpurun_out/r3b_parity.log; echo PARITY_FAILED; exit 1; }
tail -2 gpurun_out/r3b_parity.log
( for P in "valu x" "mfma asm"; do set -- $P; echo "== STARKHIP_NTT_PATH=$1 STARKHIP_MFMA_BFLY=$2";
  export STARKHIP_NTT_PATH=$1 STARKHIP_MFMA_BFLY=$2
  timeout -k 10 120 python3 tools/ntt_batch_time.py 20 1 8 32 && timeout -k 10 120 python3 tools/ntt_batch_time.py 24 1 && timeout -k 10 120 python3 tools/ntt_batch_time.py 16 64 && timeout -k 10 120 python3 tools/ntt_batch_time.py 19 64 || exit 1; done ) > gpurun_out/r3b_paths.txt 2>&1
cat gpurun_out/r3b_paths.txt
