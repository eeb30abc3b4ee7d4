#!/usr/bin/env python3
"""Generates blake2s_asm.inc: the ten rounds of the BLAKE2s compression function (RFC 7693; blake2s.cuh has the plain C++
form) as gfx950 inline-asm blocks: four columns (or diagonals) in lock-step, plain two-operand adds, and a taken branch to
the next instruction after every group of rotates.

Why (profiles/r04_blake2s_issue_rate_study.txt; microbenchmarks under lab/r04).  On gfx950 v_add_u32 / v_xor_b32 (VOP2)
issue at about 0.9 ns per wave-instruction per SIMD and v_alignbit_b32 / v_add3_u32 (VOP3) at about 1.72 ns -- but in a long
mixed straight-line stream such as this hash the fast ones cost 1.5-1.6 ns, whatever their order: runs of 24 fast instructions
recover the fast rate in a 32-instruction LOOP body and lose it again in a 224-instruction one.  The difference is the loop's
taken branch: a taken `s_branch` in front of a run of fast instructions restores their rate in straight-line code too (an
`s_nop` in the same place does not).  G is "add add xor | rot | add xor | rot | add add xor | rot | add xor | rot"; with the
four columns of a half-round in lock-step the fast runs are 12 and 8 instructions long, each started by a branch.
Bare pair-hash loop (tools/blake_occ.hip, G hashes/s, 4-8 waves per SIMD):
    compiler, C++ rounds                                   39.8-40.1
    lock-step, two adds, no branch                          35.7
    lock-step, v_add3, VOP2 ops in 8-byte encoding          43.5-44.4      (uniform instruction size helps without branches)
    lock-step, v_add3, branch after the rotates             48.0-48.3      <- emitted by default (see below)
    lock-step, two adds, branch after the rotates           50.3-51.3      (--two-adds)
    model: 800 fast + 320 slow instructions at 0.9 / 1.72 ns     51.6
    lock-step, two adds, branches, rotr 16 as two v_xor_b32_sdwa     46.8-47.4      (--sdwa16: the sub-dword forms issue at the slow rate)
Which of the last two: the bare loop runs a few milliseconds, inside the chip's boost window.  A SUSTAINED run is power-limited (the package
sits at 1.3-1.4 kW, profiles/r04_power_and_clocks.txt), and there the form with FEWER instructions wins although it issues more slowly:
ten Merkle commits of 2^24 leaves 0.587-0.601 ms against 0.604-0.614, config 5 4869-4979 against 4848-4901 proofs/s (A/B, alternating,
one session; profiles/r04_blake2s_issue_rate_study.txt).
--add3 / --two-adds / --e64 / --no-branch / --branch-after=... / --align / --sdwa16 select the other forms for A/B builds.

Each half-round (4 x G) is one asm block: 16 state registers in/out, 8 message words in (24 operands; inline asm allows 30).
The blocks are not volatile: the compiler may move whole blocks of two independent hashes past each other, never inside.

    python3 gen_blake2s_asm.py            # writes blake2s_asm.inc next to this file
"""
import os
import sys

SIGMA = [
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15],
    [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
    [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4],
    [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
    [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13],
    [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
    [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11],
    [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
    [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5],
    [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0],
]
COLS = [(0, 4, 8, 12), (1, 5, 9, 13), (2, 6, 10, 14), (3, 7, 11, 15)]
DIAGS = [(0, 5, 10, 15), (1, 6, 11, 12), (2, 7, 8, 13), (3, 4, 9, 14)]


ADD3 = True    # a = a + b + x as one v_add3_u32 (VOP3) instead of two v_add_u32 (--two-adds: the other form)
E64 = False    # the VOP2 adds / xors in their 8-byte VOP3 encoding
ALIGN = False  # every block starts 8-byte aligned (.p2align 3): with four columns per line the 8-byte instructions stay aligned
SDWA16 = False # d = rotr(d ^ a, 16) as two sub-dword xors into a scratch register (v_xor_b32_sdwa) instead of v_xor_b32 + v_alignbit_b32
BRANCH = 0     # a taken s_branch to the next instruction after every BRANCH lines of a half-round (0 = none)
BRANCH_OP = "s_branch 0"
BRANCH_AFTER = (3, 6, 9, 12)  # a taken s_branch to the next instruction after these lines (1-based) of a half-round: the rotates
# (two-adds form: lines 4, 7, 11, 14)


def half_round(groups):
    """asm text for four G's in lock-step; operands %0..%15 = v[0..15], %16+2g / %17+2g = x, y of group g."""
    lines = []
    sfx = "_e64" if E64 else ""

    def each(fmt):
        for g, (a, b, c, d) in enumerate(groups):
            lines.append(fmt.format(a=a, b=b, c=c, d=d, x=16 + 2 * g, y=17 + 2 * g))

    def add_ab(m):
        if ADD3:
            each("v_add3_u32 %{a}, %{a}, %{b}, %{" + m + "}")
        else:
            each("v_add_u32" + sfx + " %{a}, %{a}, %{b}")
            each("v_add_u32" + sfx + " %{a}, %{a}, %{" + m + "}")

    add_ab("x")
    if SDWA16:
        # operands %24..%27: scratch registers t_g; t = rotr(d ^ a, 16) word by word, d comes home with the next xor
        def each_t(fmt):
            for g, (a, b, c, d) in enumerate(groups):
                lines.append(fmt.format(a=a, b=b, c=c, d=d, t=24 + g))
        each_t("v_xor_b32_sdwa %{t}, %{d}, %{a} dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0")
        each_t("v_xor_b32_sdwa %{t}, %{d}, %{a} dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1")
        each_t("v_add_u32" + sfx + " %{c}, %{c}, %{t}")
    else:
        each("v_xor_b32" + sfx + " %{d}, %{d}, %{a}")
        each("v_alignbit_b32 %{d}, %{d}, %{d}, 16")
        each("v_add_u32" + sfx + " %{c}, %{c}, %{d}")
    each("v_xor_b32" + sfx + " %{b}, %{b}, %{c}")
    each("v_alignbit_b32 %{b}, %{b}, %{b}, 12")
    add_ab("y")
    if SDWA16:
        each_t("v_xor_b32" + sfx + " %{d}, %{t}, %{a}")
    else:
        each("v_xor_b32" + sfx + " %{d}, %{d}, %{a}")
    each("v_alignbit_b32 %{d}, %{d}, %{d}, 8")
    each("v_add_u32" + sfx + " %{c}, %{c}, %{d}")
    each("v_xor_b32" + sfx + " %{b}, %{b}, %{c}")
    each("v_alignbit_b32 %{b}, %{b}, %{b}, 7")
    if BRANCH_AFTER:
        out = []
        for i, l in enumerate(lines):
            out.append(l)
            if (i + 1) % 4 == 0 and (i + 1) // 4 in BRANCH_AFTER:
                out.append(BRANCH_OP)
        return out
    if BRANCH:
        out = []
        for i, l in enumerate(lines):
            out.append(l)
            if (i + 1) % (4 * BRANCH) == 0:
                out.append(BRANCH_OP)
        return out
    return lines


def simulate(m, v):
    """The generated operand wiring, executed in Python (checked against a plain G-by-G evaluation in main)."""
    M = 0xFFFFFFFF

    def rotr(x, n):
        return ((x >> n) | (x << (32 - n))) & M

    v = list(v)
    for r in range(10):
        for half, groups in enumerate((COLS, DIAGS)):
            ops = list(v) + [m[SIGMA[r][8 * half + i]] for i in range(8)] + [0xDEADBEEF] * 4
            for line in half_round(groups):
                op, rest = line.split(" ", 1)
                if op.startswith("s_"):
                    continue
                if op == "v_xor_b32_sdwa":
                    regs, mods = rest.split(" dst_sel:")
                    dst, s0, s1 = [int(a.strip()[1:]) for a in regs.split(",")]
                    f = dict(kv.split(":") for kv in ("dst_sel:" + mods).split())
                    pick = lambda x, sel: (x >> 16) if sel == "WORD_1" else (x & 0xFFFF)
                    val16 = pick(ops[s0], f["src0_sel"]) ^ pick(ops[s1], f["src1_sel"])
                    keep = ops[dst] if f["dst_unused"] == "UNUSED_PRESERVE" else 0
                    ops[dst] = (keep & 0x0000FFFF) | (val16 << 16) if f["dst_sel"] == "WORD_1" else (keep & 0xFFFF0000) | val16
                    continue
                args = [a.strip() for a in rest.split(",")]
                val = [ops[int(a[1:])] if a.startswith("%") else int(a) for a in args]
                dst = int(args[0][1:])
                op = op.replace("_e64", "")
                if op == "s_branch":
                    continue
                if op == "v_add_u32":
                    ops[dst] = (val[1] + val[2]) & M
                elif op == "v_add3_u32":
                    ops[dst] = (val[1] + val[2] + val[3]) & M
                elif op == "v_xor_b32":
                    ops[dst] = val[1] ^ val[2]
                elif op == "v_alignbit_b32":
                    assert args[1] == args[2]
                    ops[dst] = rotr(val[1], val[3])
                else:
                    raise ValueError(op)
            v = ops[:16]
    return v


def reference(m, v):
    M = 0xFFFFFFFF

    def rotr(x, n):
        return ((x >> n) | (x << (32 - n))) & M

    v = list(v)

    def G(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & M
        v[d] = rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & M
        v[b] = rotr(v[b] ^ v[c], 12)
        v[a] = (v[a] + v[b] + y) & M
        v[d] = rotr(v[d] ^ v[a], 8)
        v[c] = (v[c] + v[d]) & M
        v[b] = rotr(v[b] ^ v[c], 7)

    for r in range(10):
        s = SIGMA[r]
        for g, (a, b, c, d) in enumerate(COLS):
            G(a, b, c, d, m[s[2 * g]], m[s[2 * g + 1]])
        for g, (a, b, c, d) in enumerate(DIAGS):
            G(a, b, c, d, m[s[8 + 2 * g]], m[s[9 + 2 * g]])
    return v


def emit():
    out = []
    out.append("// blake2s_asm.inc -- GENERATED by gen_blake2s_asm.py; do not edit.  The ten rounds of BLAKE2s's compression function on")
    out.append("// v[0..15] with message words m[0..15], each half-round (four G's) one asm block in lock-step column order.")
    out.append("#define B2A_STATE \"+v\"(v[0]), \"+v\"(v[1]), \"+v\"(v[2]), \"+v\"(v[3]), \"+v\"(v[4]), \"+v\"(v[5]), \"+v\"(v[6]), \"+v\"(v[7]), \\")
    out.append("                  \"+v\"(v[8]), \"+v\"(v[9]), \"+v\"(v[10]), \"+v\"(v[11]), \"+v\"(v[12]), \"+v\"(v[13]), \"+v\"(v[14]), \"+v\"(v[15])")
    for half, groups in enumerate((COLS, DIAGS)):
        name = "B2A_COLS" if half == 0 else "B2A_DIAGS"
        body = " \\\n  ".join('"%s\\n"' % l for l in ([".p2align 3"] if ALIGN else []) + half_round(groups))
        out.append("#define %s \\\n  %s" % (name, body))
    out.append("#ifndef B2A_QUAL")
    out.append("#define B2A_QUAL  // -DB2A_QUAL=volatile keeps the blocks of independent hashes in program order (experiments)")
    out.append("#endif")
    out.append("__device__ __forceinline__ void b2_rounds_asm(uint32_t (&v)[16], const uint32_t (&m)[16]) {")
    if SDWA16:
        out.append("  uint32_t t0, t1, t2, t3, mm[16];")
        out.append("  for (int i = 0; i < 16; ++i) mm[i] = m[i];")
    for r in range(10):
        s = SIGMA[r]
        for half in range(2):
            ms = ", ".join('"v"(m[%d])' % s[8 * half + i] for i in range(8))
            if SDWA16:  # operands: 16 state (+v), then the 4 scratch outputs must be numbered 24..27 -> they come after the inputs in
                # the text but inline asm numbers outputs first; so the message words are passed as "+v" too (numbers 16..23)
                ms_io = ", ".join('"+v"(mm[%d])' % s[8 * half + i] for i in range(8))
                out.append("  asm B2A_QUAL(%s : B2A_STATE, %s, \"=&v\"(t0), \"=&v\"(t1), \"=&v\"(t2), \"=&v\"(t3));" %
                           ("B2A_COLS" if half == 0 else "B2A_DIAGS", ms_io))
            else:
                out.append("  asm B2A_QUAL(%s : B2A_STATE : %s);" % ("B2A_COLS" if half == 0 else "B2A_DIAGS", ms))
    out.append("}")
    out.append("#undef B2A_STATE")
    out.append("#undef B2A_COLS")
    out.append("#undef B2A_DIAGS")
    return "\n".join(out) + "\n"


def main():
    global ADD3, E64, ALIGN, BRANCH, BRANCH_OP, BRANCH_AFTER, SDWA16
    SDWA16 = "--sdwa16" in sys.argv
    import random
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--add3" in sys.argv:
        ADD3 = True
        BRANCH_AFTER = (3, 6, 9, 12)
    if "--two-adds" in sys.argv:
        ADD3 = False
        BRANCH_AFTER = (4, 7, 11, 14)
    if "--e64" in sys.argv:
        E64 = True
    if "--no-branch" in sys.argv:
        BRANCH_AFTER = ()
    ALIGN = "--align" in sys.argv
    for a in sys.argv:
        if a.startswith("--branch="):
            BRANCH = int(a.split("=")[1])
        if a.startswith("--branch-after="):
            BRANCH_AFTER = tuple(int(x) for x in a.split("=")[1].split(","))
        if a.startswith("--branch-op="):
            BRANCH_OP = a.split("=", 1)[1]
    sys.argv = sys.argv[:1] + args
    rng = random.Random(1)
    for _ in range(50):
        m = [rng.getrandbits(32) for _ in range(16)]
        v = [rng.getrandbits(32) for _ in range(16)]
        assert simulate(m, v) == reference(m, v)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "blake2s_asm.inc")
    with open(path, "w") as fh:
        fh.write(emit())


if __name__ == "__main__":
    main()
