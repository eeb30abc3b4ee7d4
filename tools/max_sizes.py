#!/usr/bin/env python3
"""Sizes above the bench's: 2^25 .. 2^30-point transforms, 2^26 / 2^28-leaf Merkle trees and the largest FRI commit the reference's
index sampling allows (domain 2^25: utils.py:69 asserts n/4 < 2^24), checked WITHOUT the oracle (it would take hours) through what the
definitions give at any size -- host arithmetic is Python ints / numpy / hashlib only:

  transform  y = NTT(x), x dense (seeded):
    (a) T = 1024 outputs exactly: y[j n/T] = sum_r w^(j (n/T) r) S_r with S_r = sum of the x_i with i = r mod T (numpy sums over the
        downloaded limbs, then a T-point DFT in Python ints);
    (b) invNTT(y) == x on every element;
    (c) K coefficients changed by d_k at positions p_k (first, last, powers of two, random): the outputs at J sampled positions
        move by exactly sum_k d_k w^(j p_k)  (linearity + the definition at arbitrary j).
  Merkle     the root and 64 sampled branches recomputed with hashlib from downloaded leaves and siblings; the whole tree against
             hashlib for --full-merkle-log (default 2^26: about a minute of host hashing with the C oracle absent on purpose).
  FRI        sh_dev_fri_prove on the 2^25 domain, the proof checked by the host verifier sh_fri_verify and by the Python verifier.

Run on the GPU box: python3 tools/max_sizes.py [--logs 25 26 28 30] > gpurun_out/max_sizes.txt
"""
import argparse
import ctypes
import hashlib
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starks_amd import _lib  # noqa: E402

P = _lib.MIMC_P
T = 1024


def root_of(n):
    return pow(7, (P - 1) // n, P)


class Dev:
    def __init__(self):
        self.L, self.ctx = _lib.lib(), _lib.ctx()

    def ck(self, rc, what):
        _lib.check(rc, what)

    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        self.ck(self.L.sh_dev_alloc(self.ctx, nbytes, ctypes.byref(p)), "alloc %d" % nbytes)
        return p

    def free(self, p):
        self.ck(self.L.sh_dev_free(self.ctx, p), "free")

    def sync(self):
        self.ck(self.L.sh_sync(self.ctx), "sync")

    def elems(self, d, idx):
        """canonical values of the elements at the given positions (one small transfer each)"""
        out = []
        b = ctypes.create_string_buffer(32)
        for i in idx:
            self.ck(self.L.sh_dev_to_wire(self.ctx, ctypes.c_void_p(d.value + 32 * i), b, 1), "to_wire")
            out.append(int.from_bytes(b.raw, "big"))
        return out

    def put(self, d, i, v):
        self.ck(self.L.sh_dev_from_wire(self.ctx, int(v).to_bytes(32, "big"), ctypes.c_void_p(d.value + 32 * i), 1), "from_wire")


def class_sums(dev, d, n, chunk_log=24):
    """S_r = sum of the elements at positions = r mod T, from the raw limbs (8 x u32, little-endian limbs, lazily reduced: the sums
    are taken over the integers and reduced at the end)."""
    ch = min(n, 1 << chunk_log)
    buf = np.empty(ch * 8, dtype="<u4")
    acc = np.zeros((T, 8), dtype=object)
    for off in range(0, n, ch):
        dev.ck(dev.L.sh_dev_download(dev.ctx, ctypes.c_void_p(d.value + 32 * off), buf.ctypes.data_as(ctypes.c_void_p), 32 * ch), "dl")
        acc += buf.reshape(ch // T, T, 8).sum(axis=0, dtype=np.uint64).astype(object)
    return [sum(int(acc[r, k]) << (32 * k) for k in range(8)) % P for r in range(T)]


def equal_everywhere(dev, da, db, n, chunk_log=22):
    ch = min(n, 1 << chunk_log)
    a, b = ctypes.create_string_buffer(32 * ch), ctypes.create_string_buffer(32 * ch)
    for off in range(0, n, ch):
        dev.ck(dev.L.sh_dev_to_wire(dev.ctx, ctypes.c_void_p(da.value + 32 * off), a, ch), "dl")
        dev.ck(dev.L.sh_dev_to_wire(dev.ctx, ctypes.c_void_p(db.value + 32 * off), b, ch), "dl")
        if a.raw != b.raw:
            return False
    return True


def check_transform(dev, logn, rng):
    n = 1 << logn
    w = root_of(n)
    wb = w.to_bytes(32, "big")
    t0 = time.time()
    dx, dy = dev.alloc(32 * n), dev.alloc(32 * n)
    dev.ck(dev.L.sh_dev_fill_seeded(dev.ctx, dx, n, 0x5eed), "fill")
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dx, dy, n, 1, wb, 0), "ntt")
    dev.sync()
    t1 = time.time()
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dx, dy, n, 1, wb, 0), "ntt")
    dev.sync()
    t_fwd = time.time() - t1
    # (a) T outputs from the residue-class sums
    S = class_sums(dev, dx, n)
    g = pow(w, n // T, P)  # order T
    gp = [pow(g, e, P) for e in range(T)]
    want = [sum(S[r] * gp[(j * r) % T] for r in range(T)) % P for j in range(T)]
    got = dev.elems(dy, [j * (n // T) for j in range(T)])
    ok_a = want == got
    # (c) sampled outputs before the change ...
    pos = sorted(set([0, 1, n - 1, n // 2, n // 2 + 1] + [1 << k for k in range(2, logn, 3)] + [rng.randrange(n) for _ in range(24)]))
    J = sorted(set([0, 1, n - 1, n // 2, n // 4 + 1] + [(1 << k) + 1 for k in range(2, logn, 3)] + [rng.randrange(n) for _ in range(96)]))
    y_before = dev.elems(dy, J)
    # (b) the inverse, in place on y, against x
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dy, dy, n, 1, wb, 1), "intt")
    ok_b = equal_everywhere(dev, dx, dy, n)
    # ... and after it
    old = dev.elems(dx, pos)
    delta = [rng.randrange(1, P) for _ in pos]
    for p_, o, d in zip(pos, old, delta):
        dev.put(dx, p_, (o + d) % P)
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dx, dy, n, 1, wb, 0), "ntt")
    y_after = dev.elems(dy, J)
    ok_c = all((ya - yb) % P == sum(d * pow(w, (j * p_) % n, P) for p_, d in zip(pos, delta)) % P
               for j, yb, ya in zip(J, y_before, y_after))
    passes = int(dev.L.sh_ntt_passes(n, 1))
    dev.free(dx)
    dev.free(dy)
    print("ntt 2^%d: %d passes, forward %.1f ms (%.2f G elements/s) | %d outputs from class sums %s | inverse == input on every element %s"
          " | %d changed coefficients move %d sampled outputs as the definition says %s | %.0f s"
          % (logn, passes, t_fwd * 1e3, n / t_fwd / 1e9, T, ok_a, ok_b, len(pos), len(J), ok_c, time.time() - t0), flush=True)
    return ok_a and ok_b and ok_c


def seeded(seed, i):
    """x_i = BLAKE2s(seed_le64 || i_le64) mod p (what sh_dev_fill_seeded writes)"""
    return int.from_bytes(hashlib.blake2s(seed.to_bytes(8, "little") + i.to_bytes(8, "little")).digest(), "big") % P


def check_transform_in_place(dev, logn, rng):
    """2^32 points = 128 GiB: the vector and the library's work buffer are all that fits, so x is not kept -- every transform runs in
    place, (b) becomes: after the inverse, the T residue-class sums are those of x again (every element counts) and 256 sampled
    elements are the seeded values."""
    n = 1 << logn
    w = root_of(n)
    wb = w.to_bytes(32, "big")
    t0 = time.time()
    dx = dev.alloc(32 * n)
    dev.ck(dev.L.sh_dev_fill_seeded(dev.ctx, dx, n, 0x5eed), "fill")
    I = sorted(set([0, 1, n - 1, n // 2, (1 << 31) - 1 if n > (1 << 31) else 5, min(n - 1, 1 << 31)] + [rng.randrange(n) for _ in range(250)]))
    ok_fill = dev.elems(dx, I) == [seeded(0x5eed, i) for i in I]
    S = class_sums(dev, dx, n)
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dx, dx, n, 1, wb, 0), "ntt")   # untimed: builds both plans and the work buffer
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dx, dx, n, 1, wb, 1), "intt")
    dev.sync()
    t1 = time.time()
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dx, dx, n, 1, wb, 0), "ntt")
    dev.sync()
    t_fwd = time.time() - t1
    g = pow(w, n // T, P)
    gp = [pow(g, e, P) for e in range(T)]
    want = [sum(S[r] * gp[(j * r) % T] for r in range(T)) % P for j in range(T)]
    ok_a = want == dev.elems(dx, [j * (n // T) for j in range(T)])
    pos = sorted(set([0, 1, n - 1, n // 2, n // 2 + 1] + [1 << k for k in range(2, logn, 3)] + [rng.randrange(n) for _ in range(24)]))
    J = sorted(set([0, 1, n - 1, n // 2, n // 4 + 1] + [(1 << k) + 1 for k in range(2, logn, 3)] + [rng.randrange(n) for _ in range(96)]))
    y_before = dev.elems(dx, J)
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dx, dx, n, 1, wb, 1), "intt")
    ok_b = dev.elems(dx, I) == [seeded(0x5eed, i) for i in I] and class_sums(dev, dx, n) == S
    delta = [rng.randrange(1, P) for _ in pos]
    for p_, d in zip(pos, delta):
        dev.put(dx, p_, (seeded(0x5eed, p_) + d) % P)
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dx, dx, n, 1, wb, 0), "ntt")
    y_after = dev.elems(dx, J)
    ok_c = all((ya - yb) % P == sum(d * pow(w, (j * p_) % n, P) for p_, d in zip(pos, delta)) % P
               for j, yb, ya in zip(J, y_before, y_after))
    dev.free(dx)
    print("ntt 2^%d in place (%d GiB): forward %.1f ms (%.2f G elements/s) | %d sampled inputs are the seeded values %s | %d outputs "
          "from class sums %s | after the inverse: sampled elements and all class sums are the input's %s | %d changed coefficients "
          "move %d sampled outputs as the definition says %s | %.0f s"
          % (logn, (32 * n) >> 30, t_fwd * 1e3, n / t_fwd / 1e9, len(I), ok_fill, T, ok_a, ok_b, len(pos), len(J), ok_c,
             time.time() - t0), flush=True)
    return ok_fill and ok_a and ok_b and ok_c


def check_host_api(dev, logn):
    """sh_ntt from and to caller (pageable) buffers of more than 4 GiB: staged through the pinned slots, the same bytes as the device
    API's result"""
    n = 1 << logn
    wb = root_of(n).to_bytes(32, "big")
    t0 = time.time()
    dx, dy = dev.alloc(32 * n), dev.alloc(32 * n)
    dev.ck(dev.L.sh_dev_fill_seeded(dev.ctx, dx, n, 0xabcd), "fill")
    dev.ck(dev.L.sh_dev_ntt(dev.ctx, dx, dy, n, 1, wb, 0), "ntt")
    x, y, y2 = (ctypes.create_string_buffer(32 * n) for _ in range(3))
    ch = 1 << 24
    for off in range(0, n, ch):
        for d, h in ((dx, x), (dy, y)):
            dev.ck(dev.L.sh_dev_to_wire(dev.ctx, ctypes.c_void_p(d.value + 32 * off),
                                        ctypes.cast(ctypes.addressof(h) + 32 * off, ctypes.c_void_p), min(ch, n - off)), "dl")
    dev.free(dx)
    dev.free(dy)
    t1 = time.time()
    dev.ck(dev.L.sh_ntt(dev.ctx, x, n, y2, n, wb, 0), "sh_ntt")
    t_h = time.time() - t1
    same = hashlib.sha256(y).digest() == hashlib.sha256(y2).digest()
    print("sh_ntt from / to pageable host buffers, 2^%d points (%d GiB each way): %.2f s, same bytes as the device API %s | %.0f s"
          % (logn, (32 * n) >> 30, t_h, same, time.time() - t0), flush=True)
    return same


def blake(b):
    return hashlib.blake2s(b).digest()


def check_merkle(dev, logn, rng, full):
    n = 1 << logn
    t0 = time.time()
    dx, dn = dev.alloc(32 * n), dev.alloc(64 * n)
    dev.ck(dev.L.sh_dev_fill_seeded(dev.ctx, dx, n, 0x77), "fill")
    dev.ck(dev.L.sh_dev_merkelize(dev.ctx, dx, n, 1, dn), "merkelize")
    dev.sync()
    t1 = time.time()
    dev.ck(dev.L.sh_dev_merkelize(dev.ctx, dx, n, 1, dn), "merkelize")
    dev.sync()
    t_m = time.time() - t1

    def node(i):
        b = ctypes.create_string_buffer(32)
        dev.ck(dev.L.sh_dev_download(dev.ctx, ctypes.c_void_p(dn.value + 32 * i), b, 32), "dl")
        return b.raw

    root = node(1)
    ok = True
    q = n // 4
    for idx in [0, 1, n - 1, q, q - 1, 3 * q + 5] + [rng.randrange(n) for _ in range(58)]:
        pi = idx // q + 4 * (idx % q)  # get_index_in_permuted (merkle_tree.py:26-33)
        v = dev.elems(dx, [idx])[0].to_bytes(32, "big")
        i = n + pi
        ok = ok and node(i) == v
        while i > 1:
            sib = node(i ^ 1)
            v = blake(v + sib) if i % 2 == 0 else blake(sib + v)
            i //= 2
        ok = ok and v == root
    msg = ""
    if full:
        # the whole tree with hashlib: leaves in permute4 order, then level by level
        wire = ctypes.create_string_buffer(32 * n)
        ch = 1 << 22
        for off in range(0, n, ch):
            dev.ck(dev.L.sh_dev_to_wire(dev.ctx, ctypes.c_void_p(dx.value + 32 * off),
                                        ctypes.cast(ctypes.addressof(wire) + 32 * off, ctypes.c_void_p), ch), "dl")
        a = np.frombuffer(wire, dtype=np.uint8).reshape(4, q, 32).transpose(1, 0, 2).reshape(n, 32)  # out[4 i + j] = in[i + j q]
        level = a.tobytes()
        del a, wire
        m = n
        while m > 1:
            level = b"".join(blake(level[64 * i:64 * i + 64]) for i in range(m // 2))
            m //= 2
        full_ok = level == root
        ok = ok and full_ok
        msg = " | root == hashlib over all %d leaves %s" % (n, full_ok)
    dev.free(dx)
    dev.free(dn)
    print("merkle 2^%d: %.2f ms | 64 branches (leaf, siblings, root) recomputed with hashlib %s%s | %.0f s"
          % (logn, t_m * 1e3, ok, msg, time.time() - t0), flush=True)
    return ok


def check_fri_max(dev, logsteps=22):
    """the largest commit utils.py:69 allows: a 2^22-step trace, domain 2^25 (its first column has 2^23 < 2^24 entries)"""
    from starks_amd import fri as sfri
    L = dev.L
    steps, ext = 1 << logsteps, 8
    n = steps * ext
    g2 = root_of(n)
    wb = g2.to_bytes(32, "big")
    t0 = time.time()
    dc, dv, dt = dev.alloc(32 * n), dev.alloc(32 * n), dev.alloc(64 * n)
    dev.ck(L.sh_dev_fill_seeded(dev.ctx, dc, steps, 0xf1), "fill")  # a polynomial of degree < steps
    plen = sfri.proof_len(n, steps, 40)
    dp = dev.alloc(plen)
    flat = ctypes.create_string_buffer(plen)
    dev.ck(L.sh_dev_fri_prove_coeffs(dev.ctx, dc, steps, n, wb, steps, ext, 40, 1, dp), "fri_prove_coeffs")
    dev.sync()
    t1 = time.time()
    dev.ck(L.sh_dev_fri_prove_coeffs(dev.ctx, dc, steps, n, wb, steps, ext, 40, 1, dp), "fri_prove_coeffs")
    dev.sync()
    t_p = time.time() - t1
    dev.ck(L.sh_dev_download(dev.ctx, dp, flat, plen), "dl")
    # the same bytes from the zero-padded vector; then the commitment the verifier starts from
    zeros = bytes(32 << 20)
    for off in range(steps, n, 1 << 20):
        dev.ck(L.sh_dev_upload(dev.ctx, zeros, ctypes.c_void_p(dc.value + 32 * off), len(zeros)), "ul")
    dev.ck(L.sh_dev_fri_prove(dev.ctx, dc, n, wb, steps, ext, 40, 1, dp), "fri_prove")
    padded = ctypes.create_string_buffer(plen)
    dev.ck(L.sh_dev_download(dev.ctx, dp, padded, plen), "dl")
    same = padded.raw == flat.raw
    dev.ck(L.sh_dev_ntt(dev.ctx, dc, dv, n, 1, wb, 0), "ntt")
    dev.ck(L.sh_dev_merkelize(dev.ctx, dv, n, 1, dt), "merkelize")
    root = ctypes.create_string_buffer(64)
    dev.ck(L.sh_dev_download(dev.ctx, dt, root, 64), "dl")
    mroot = root.raw[32:64]
    native = bool(sfri.verify_flat(flat.raw, mroot, n, g2, steps, ext, 40))
    py = bool(sfri.verify_low_degree_proof(sfri.unpack_proof(flat.raw, n, steps, 40), mroot, g2, steps, ext))
    tampered = bytearray(flat.raw)
    tampered[len(tampered) // 2] ^= 1
    try:
        sfri.verify_flat(bytes(tampered), mroot, n, g2, steps, ext, 40)
        rejects = False
    except AssertionError:
        rejects = True
    for d in (dc, dv, dt, dp):
        dev.free(d)
    print("fri commit, 2^%d-step trace (domain 2^%d, the largest utils.py:69 allows): %.2f ms, proof %d bytes | same bytes from the "
          "padded vector %s | sh_fri_verify accepts %s | Python verifier accepts %s | a flipped bit is rejected %s | %.0f s"
          % (logsteps, logsteps + 3, t_p * 1e3, plen, same, native, py, rejects, time.time() - t0), flush=True)
    return same and native and py and rejects


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logs", type=int, nargs="*", default=[25, 26, 27, 28, 30])
    ap.add_argument("--merkle-logs", type=int, nargs="*", default=[25, 26, 28])
    ap.add_argument("--full-merkle-log", type=int, default=25)
    ap.add_argument("--in-place-logs", type=int, nargs="*", default=[], help="e.g. 32: the vector alone is 128 GiB")
    ap.add_argument("--host-logs", type=int, nargs="*", default=[], help="e.g. 28: sh_ntt on 8 GiB host buffers")
    ap.add_argument("--no-fri", action="store_true")
    ap.add_argument("--seed", type=int, default=5)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    dev = Dev()
    ok = True
    for logn in args.logs:
        ok = check_transform(dev, logn, rng) and ok
    for logn in args.in_place_logs:
        ok = check_transform_in_place(dev, logn, rng) and ok
    for logn in args.merkle_logs:
        ok = check_merkle(dev, logn, rng, full=logn <= args.full_merkle_log) and ok
    if not args.no_fri:
        ok = check_fri_max(dev) and ok
    for logn in args.host_logs:
        ok = check_host_api(dev, logn) and ok
    print("ALL OK" if ok else "MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
