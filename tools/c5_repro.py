#!/usr/bin/env python3
"""Determinism check of the many-proof workload (config 5): the 512 proofs proved through 1 and through 2 library contexts, several
times over, compared byte for byte with the first single-context run; a difference is reported by unit, offset and proof region.

    python tools/c5_repro.py [--iters 8] [--streams 2] [--units 512] [--chunk 128] [--ntt-first]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def download_all(dev, sh):
    k = len(sh.units)
    out = ctypes.create_string_buffer(sh.plen * k)
    dev.ck(dev.L.sh_dev_download(dev.ctx, sh.d_proofs, out, sh.plen * k), "dl")
    return out.raw


def regions(steps, ext, degree, samples=80):
    """(name, begin, end) byte ranges of one flat proof: header + spot checks, then every FRI round, then the final values."""
    n = steps * ext
    lg = n.bit_length() - 1
    k = 6
    head = 64 + samples * 32 * (2 * (2 * k + (lg - 1)) + (lg + 1))
    out = [("stark header+spot checks", 0, head)]
    off, nn, md, r = head, n, steps * degree, 0
    while md > 16:
        l = nn.bit_length() - 1
        size = 32 + 40 * 32 * ((l - 1) + 4 * (l + 1))
        out.append(("fri round %d (n=2^%d)" % (r, l), off, off + size))
        off += size
        nn >>= 2
        md >>= 2
        r += 1
    out.append(("fri final values", off, off + 32 * nn))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--units", type=int, default=512)
    ap.add_argument("--chunk", type=int, default=128)
    ap.add_argument("--logsteps", type=int, default=16)
    ap.add_argument("--quiet", action="store_true", help="one line per iteration")
    ap.add_argument("--ntt-first", action="store_true", help="run a 2^24-point NTT pair first, as the default bench run does")
    args = ap.parse_args()
    os.environ.setdefault("STARKHIP_DEVICE", "0")
    dev = bench.Dev()
    L, ctx = dev.L, dev.ctx
    if args.ntt_first:
        n = 1 << 24
        w = bench.root_of(n).to_bytes(32, "big")
        dx, dy = dev.alloc(32 * n), dev.alloc(32 * n)
        dev.ck(L.sh_dev_fill_seeded(ctx, dx, n, 1), "fill")
        for _ in range(3):
            dev.ck(L.sh_dev_ntt(ctx, dx, dy, n, 1, w, 0), "ntt")
            dev.ck(L.sh_dev_ntt(ctx, dy, dy, n, 1, w, 1), "intt")
        dev.sync()
        dev.free(dx)
        dev.free(dy)
    steps = 1 << args.logsteps
    units = list(range(args.units))
    sh1 = bench.ProofShard(dev, units, steps, 8, args.chunk, 1)
    sh1.prove_all()
    sh1.headers()
    ref = download_all(dev, sh1)
    sh1.prove_all()
    sh1.headers()
    again = download_all(dev, sh1)
    print("single context, second run equal to the first:", again == ref, flush=True)
    regs = regions(steps, 8, sh1.degree)
    assert regs[-1][2] == sh1.plen, (regs[-1], sh1.plen)
    plen = sh1.plen
    sh1.close()
    sh = bench.ProofShard(dev, units, steps, 8, args.chunk, args.streams)
    bad_total = 0
    for it in range(args.iters):
        sh.prove_all()
        sh.headers()
        got = download_all(dev, sh)
        if got == ref:
            print("iter %d: all %d proofs equal" % (it, len(units)), flush=True)
            continue
        if args.quiet:
            bad = [u for u in range(len(units)) if ref[u * plen:(u + 1) * plen] != got[u * plen:(u + 1) * plen]]
            whole = sum(1 for u in bad if ref[u * plen:u * plen + 32] != got[u * plen:u * plen + 32])
            bad_total += len(bad)
            print("iter %d: %d proofs differ (%d from the m_root on), per launch %s" %
                  (it, len(bad), whole, [sum(1 for u in bad if u // args.chunk == l) for l in range((len(units) + args.chunk - 1) // args.chunk)]),
                  flush=True)
            continue
        for u in range(len(units)):
            a, b = ref[u * plen:(u + 1) * plen], got[u * plen:(u + 1) * plen]
            if a == b:
                continue
            bad_total += 1
            diffs = [i for i in range(0, plen, 32) if a[i:i + 32] != b[i:i + 32]]
            where = {}
            for d in diffs:
                for name, lo, hi in regs:
                    if lo <= d < hi:
                        where.setdefault(name, []).append((d - lo) // 32)
            print("iter %d: unit %d (slot %d of its launch, launch %d) differs in %d 32-byte words:" %
                  (it, u, u % args.chunk, u // args.chunk, len(diffs)), flush=True)
            for name, lst in where.items():
                print("     %-28s %5d words, first at word %d, last at word %d" % (name, len(lst), lst[0], lst[-1]), flush=True)
    print("TOTAL differing proofs over %d iterations: %d" % (args.iters, bad_total))
    sh.close()


if __name__ == "__main__":
    main()
