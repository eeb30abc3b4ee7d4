echo "add3, e32; s_branch after lines: p1 = 2,5,8,11 (after the xors)  p2 = 1,3,7,9 (before each fast run)  p3 = every transition  p4 = 3,9  p5 = 1,3,4,6,7,9,10,12  p6 = 3,6,9,12 (after the rotates); two adds, e32: p7 = around the rotates  p8 = after the rotates"
for n in p1 p2 p3 p4 p5 p6 p7 p8; do echo "== $n"; timeout -k 10 60 tools/r04/occ_$n | grep "ILP [12], [48] waves"; done
