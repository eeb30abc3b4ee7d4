for n in basealign add3align add3e64align add3e64; do echo "== $n"; timeout -k 10 60 tools/r04/occ_$n | grep "ILP [12], [48] waves"; done
