cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export STARKHIP_LIB=$PWD/starks_amd/libstarkhip_stamps.so STARKHIP_NTT_PATH=hybrid
for M in valu mfma; do
for d in 0 1; do echo "== hybrid mapping, shared groups' math on $M: 2^20 x 8 vectors, pass $d"; STARKHIP_HYBRID_MATH=$M STARKHIP_STAMP_PASS=$d timeout -k 10 100 python3 tools/lds_phases.py 20 8 || exit 1; done
done
