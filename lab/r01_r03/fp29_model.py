#!/usr/bin/env python3
"""Bit-accurate Python model of the unsaturated 9 x 29-bit arithmetic used inside the NTT tile kernels
(tools/fp29.cuh).  Every intermediate is asserted to fit the register width the kernel uses
(int32 limbs, int64 columns); run it to re-validate the bounds after any change to the algorithm."""
import random

P = 2**256 - 2**32 * 351 + 1
M = (1 << 29) - 1
K1, K0 = 89856, -32  # 2^261 = 32 * 2^256 == 32c = 89856 * 2^29 - 32  (mod p)
assert (1 << 261) % P == (K1 * (1 << 29) + K0) % P
I64 = 1 << 63
I32 = 1 << 31


def fits64(v):
    assert -I64 <= v < I64, v
    return v


def fits32(v):
    assert -I32 <= v < I32, v
    return v


def val(l):
    return sum(x << (29 * i) for i, x in enumerate(l))


def from_int_unsigned(x):
    """8x32 saturated value (< 2^256) -> 9 unsigned limbs in [0, 2^29)."""
    return [(x >> (29 * i)) & M for i in range(9)]


def to_balanced(x):
    """canonical integer -> balanced digits in [-2^28, 2^28) (twiddle tables)."""
    out, c = [], 0
    for i in range(9):
        d = ((x >> (29 * i)) & M) + c
        c = 0
        if d >= 1 << 28:
            d -= 1 << 29
            c = 1
        out.append(d)
    assert c == 0 and val(out) == x
    return out


def mul(x, w):
    """x: 9 signed limbs |x_i| < 2^31 ; w: balanced twiddle |w_j| <= 2^28.
    returns 9 limbs, limbs 0..7 in [-2^28, 2^28), limb 8 small; value == x*w (mod p)."""
    BIAS = 1 << 28
    C = [0] * 18
    for k in range(17):
        acc = BIAS if k < 9 else 0
        for i in range(9):
            j = k - i
            if 0 <= j < 9:
                acc = fits64(acc + fits32(x[i]) * w[j])
        C[k] = acc
    # normalise the high half (columns 9..16) to unsigned 29-bit digits h_0..h_8 and a small signed h_9
    h = []
    for k in range(9, 17):
        h.append(C[k] & M)
        C[k + 1] = fits64(C[k + 1] + (C[k] >> 29))
    h.append(C[17] & M)
    h.append(C[17] >> 29)
    assert abs(h[9]) < 64
    # fold: 2^261 * H == H * (K1 * 2^29 + K0)
    L = C[:9] + [BIAS, BIAS]
    for j in range(10):
        L[j] = fits64(L[j] + K0 * h[j])
        L[j + 1] = fits64(L[j + 1] + K1 * h[j])
    # low pass: balanced digits (columns carry a +2^28 bias)
    r = [0] * 11
    for k in range(10):
        r[k] = (L[k] & M) - BIAS
        L[k + 1] = fits64(L[k + 1] + (L[k] >> 29))
    o0 = r[9]                 # digit of weight 2^261
    o1 = L[10] - BIAS         # weight 2^290  (small: |o1| < 2^23)
    assert abs(o1) < 1 << 24
    # second fold of O = o0 + 2^29 o1 into limbs 0..2, then ripple
    t0 = r[0] + K0 * o0
    t1 = r[1] + K1 * o0 + K0 * o1
    t2 = r[2] + K1 * o1
    fits64(t0), fits64(t1), fits64(t2)
    out = r[:9]
    c = 0
    for k, t in enumerate((t0, t1, t2)):
        t += c + BIAS
        out[k] = (t & M) - BIAS
        c = t >> 29
    out[3] = fits32(out[3] + c)  # |c| small: limb 3 stays within 2^28 + 2^18
    assert abs(c) < 1 << 18, c
    for k in range(8):
        assert abs(out[k]) <= (1 << 28) + (1 << 18)
    assert abs(out[8]) <= (1 << 28)
    return out


def add(a, b):
    return [fits32(x + y) for x, y in zip(a, b)]


def sub(a, b):
    return [fits32(x - y) for x, y in zip(a, b)]


def normalize_balanced(a):
    """limbs -> balanced digits, top limb keeps the carry (used once per pass inside the tile)."""
    out, c = [], 0
    BIAS = 1 << 28
    for k in range(8):
        t = a[k] + c + BIAS
        out.append((t & M) - BIAS)
        c = t >> 29
    out.append(fits32(a[8] + c))
    return out


PBIAS = 512 * P  # |value| of a lazy element stays < 2^264.2 for up to 16 levels of a +- (product)
P64 = [(PBIAS >> (29 * i)) & M for i in range(8)] + [PBIAS >> 232]  # top limb ~2^33: added in 64-bit


def to_saturated(a):
    """signed lazy limbs (|value| < 2^262) -> integer in [0, 2^256) congruent mod p (what gets stored)."""
    t = [x + y for x, y in zip(a, P64)]          # + 512p: value now positive, < 2^266
    c = 0
    r = []
    for k in range(8):
        v = t[k] + c
        r.append(v & M)
        c = v >> 29
    top = t[8] + c
    assert 0 <= top < 1 << 34, top               # held in 64 bits in the kernel
    r.append(top)
    v = val(r)
    assert 0 <= v < 1 << 266
    hi = r[8] >> 24                                # bits >= 256
    r[8] &= (1 << 24) - 1
    # + hi * c, c = 351*2^32 - 1  ->  limb0 += -hi, bit 32+... : add hi*351 at bit 32 = limb 1 bit 3
    lo = val(r) + hi * (351 * 2**32 - 1)
    hi2 = lo >> 256
    lo = (lo & ((1 << 256) - 1)) + hi2 * (351 * 2**32 - 1)
    assert lo < 1 << 256
    return lo


if __name__ == "__main__":
    rng = random.Random(1)
    for it in range(20000):
        if it % 4 == 0:
            x = [rng.choice([-(1 << 31), (1 << 31) - 1, rng.randrange(-(1 << 31), 1 << 31)]) for _ in range(9)]
        else:
            x = [rng.randrange(-(1 << 31), 1 << 31) for _ in range(9)]
        wv = rng.choice([0, 1, P - 1, rng.randrange(P), (1 << 255) + 12345, sum(((1 << 28) - 1) << (29 * i) for i in range(8))])
        w = to_balanced(wv % P)
        r = mul(x, w)
        assert val(r) % P == val(x) * wv % P
        s = to_saturated(r)
        assert s % P == val(x) * wv % P
    # DIT growth: 7 levels of a +- t with t a fresh product stay inside int32 and the product bound
    for it in range(300):
        a = from_int_unsigned(rng.randrange(1 << 256))
        for lv in range(7):
            b = mul([rng.randrange(-(1 << 30), 1 << 30) for _ in range(9)], to_balanced(rng.randrange(P)))
            a = add(a, b) if rng.random() < 0.5 else sub(a, b)
        n = normalize_balanced(a)
        assert val(n) == val(a)
        assert val(a) % P == to_saturated(a) % P
    print("fp29 model ok")
