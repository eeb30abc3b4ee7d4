"""TEST INFRASTRUCTURE ONLY: one host core of bench.py's cpu_baseline leg.

    python -m oracle.cpu_worker LOGN VECTOR BUDGET_S [LOG_STRIDE OFFSET]

runs forward+inverse NTTs of vector VECTOR of the bench workload (x_i = BLAKE2s(seed_le64 || i_le64) mod p)
with the C oracle (oracle/oracle.c) until BUDGET_S seconds are spent and prints one JSON line.  With LOG_STRIDE the
transform is one branch of the reference's recursion (fft.py:303-314) on the 2^(LOGN+LOG_STRIDE)-point vector 0: the
2^LOGN inputs x[OFFSET + i * 2^LOG_STRIDE] over the root w^(2^LOG_STRIDE) -- a bounded sample of a transform too long to
run whole inside the benchmark."""
import hashlib
import json
import struct
import sys
import time

from . import coracle
from .pyoracle import MIMC_P as P


def main(logn, b, budget_s, log_stride=0, offset=0):
    n = 1 << logn
    data = b"".join(hashlib.blake2s(struct.pack("<QQ", 0x5eed, b * (n << log_stride) + offset + (i << log_stride))).digest()
                    for i in range(n))
    w = pow(7, (P - 1) // n, P)  # == (the root of order n * 2^log_stride)^(2^log_stride)
    t0 = time.time()
    reps = 0
    while True:
        f = coracle.fft_bytes(data, n, w)
        back = coracle.fft_bytes(f, n, w, inverse=True)
        reps += 1
        if time.time() - t0 > budget_s or reps >= 8:
            break
    dt = time.time() - t0
    ok = all(int.from_bytes(back[32 * i:32 * i + 32], "big") == int.from_bytes(data[32 * i:32 * i + 32], "big") % P
             for i in (0, 1, n // 2, n - 1))
    print(json.dumps({"reps": reps, "seconds": dt, "fwd_sha256": hashlib.sha256(f).hexdigest(), "roundtrip_ok": ok}))


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), *[int(v) for v in sys.argv[4:6]])
