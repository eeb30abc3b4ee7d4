# tools/r04/c5_repro.py over library builds (profiles/r04_two_context_race.txt): the default library, then the diagnostic variants
# given as arguments (suffixes of starks_amd/libstarkhip<suffix>.so), then the default under AMD_SERIALIZE_KERNEL=3
for v in "" "$@"; do
  echo "=== lib$v"
  STARKHIP_LIB=$PWD/starks_amd/libstarkhip$v.so timeout -k 10 120 python tools/r04/c5_repro.py --iters 5 --quiet 2>&1 | tail -8
done
