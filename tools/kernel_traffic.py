#!/usr/bin/env python3
"""per-kernel HBM bytes and durations from three rocprofv3 runs of the same command (kernel stats, --pmc FETCH_SIZE, --pmc
WRITE_SIZE): args STATS_DIR FETCH_DIR WRITE_DIR.  HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE KiB (gfx950 correction, as in
make_traffic.py)."""
import sys
from make_traffic import per_kernel, durations
sdir, fdir, wdir = sys.argv[1:4]
F, W, D = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE"), durations(sdir)
for k in sorted(F, key=lambda k: -D.get(k, (0, 0))[1]):
    if "ntt_" not in k:
        continue
    fb = 2 * 1024 * sum(F[k]) / len(F[k])
    wb = 1024 * sum(W[k]) / len(W[k])
    calls, ns = D.get(k, (0, 0.0))
    name = k.split("(")[0]
    print("%-60s %4d launches  %8.1f us  fetch %7.1f MB  write %7.1f MB  -> %5.2f TB/s" %
          (name[-60:], calls, ns / 1e3, fb / 1e6, wb / 1e6, (fb + wb) / ns / 1e3 if ns else 0))
