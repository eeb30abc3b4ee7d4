# phase stamps of the hybrid tile pass in the steady state (workgroups 2048..3071 of 4096), matrix-core blocks vs VALU butterflies
# in the shared groups
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_stamps.so STARKHIP_NTT_PATH=hybrid STARKHIP_STAMP_BASE=2048
for M in valu mfma frag0; do
for d in 0 1; do echo "== hybrid mapping, shared groups' math: $M; 2^20 x 8 vectors, pass $d, workgroups 2048.."; STARKHIP_HYBRID_MATH=$M STARKHIP_STAMP_PASS=$d timeout -k 10 100 python3 tools/lds_phases.py 20 8 || exit 1; done
done
