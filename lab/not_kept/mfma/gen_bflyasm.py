#!/usr/bin/env python3
"""Generates mfma_bfly.inc: the butterfly stages of the matrix-core tile pass (ntt_mfma.hip) as hand-scheduled gfx950
inline asm, one block per stage.

Why asm.  One butterfly of a tile transform is (a, b) -> (a + b, (a - b) * w) with the product on the matrix cores
(mfma_tw.cuh).  From C++ the compiler emits ~120 VALU instructions plus ~105 `s_nop`s per butterfly: gfx950 needs two
wait states between a VALU write of a carry (VCC / SGPR pair) and the VALU that consumes it, 12 between an MFMA and the
first VALU read of its result, two between a VALU write and a v_permlane32_swap of the same register -- and a butterfly is
a handful of serial carry chains.  With two waves per SIMD (the register tile allows no more) those slots are not hidden.
Here every butterfly is split into streams --

    P(j)  operand preparation: TA = a ^ 0x80.., a += b (lazily reduced), b ^= 0x80.., 8 half-swaps
    M(j)  four v_mfma_i32_32x32x32_i8 (two column groups x (W * a, -W * b)), accumulators start from the offset block
    N(j)  normalisation: 2 x 16 sums at byte spacing -> 2 x (4 limbs + carry), two interleaved mad chains
    J(j)  5 half-swaps, join of the halves, fold of the top carry (2^256 == c), written over b

-- and a list scheduler interleaves the streams of successive butterflies (oldest first, younger ones fill the slots a
hazard would leave empty; every carry chain has its own SGPR pair), so the stage issues with (almost) no `s_nop`.
Hazard distances are the ones hipcc's own hazard recogniser applies on gfx950 (observed in its output: DESIGN.md
section 5).  The rare carry continuations (probability 2^-23 .. 2^-41 per lane) are out-of-line blocks behind scalar
branches.

The same module contains a functional simulator of the instruction subset (64 lanes, MFMA operand layout as established on
the hardware: mfma_tw.cuh) -- `python3 gen_bflyasm.py --selftest` runs every generated stage on random and on crafted
(rare-carry) inputs and compares with big-integer butterflies; tests/test_host_cpu.py runs it too.  Register numbers are
physical: the 16 elements of a thread are pinned to v[128:255] by the asm constraints, temporaries are clobbers.
"""
import os
import random
import sys

P = 2**256 - 351 * 2**32 + 1
M32 = 0xffffffff

# ---- physical registers -------------------------------------------------------------------------------------------------
DATA0 = 128                      # x[m] = v[128 + 8 m .. 135 + 8 m]
ACC1, ACC2 = 0, 16               # MFMA accumulators (16 each)
TA = 32                          # a ^ 0x80808080, then its swapped halves (MFMA operands a1 = TA[0:3], a2 = TA[4:7])
FRAG = 40                        # W1 40..43, N1 44..47, W2 48..51, N2 52..55
EZ1, EZ2 = 56, 58                # (e, 0) pairs: 64-bit addend of the first mad of a limb
T1, T2 = 60, 62                  # mad accumulators (pairs)
JT, JD0, JD1, JM = 64, 65, 66, 67
SM = 68
AM = 69                          # borrow mask of the add/sub butterfly's difference chain
FRAG_B = 70                      # second fragment buffer 70..85: the loads of butterfly j + 1 are issued a whole butterfly early
NTEMP = 86                       # v0 .. v85 are clobbered
# scalar scratch (clobbered)
S_CA, S_CB, S_CJ = 64, 66, 68    # carry pairs: sum chain, (spare), join chain
S_AD1, S_AD2 = 70, 72            # TwMat addresses of the two column groups
S_K80 = 74
S_SH8, S_SH16, S_SH24 = 75, 76, 77
S_351 = 78
S_BASE = 80                      # 4 x 64-bit per-level bases: s[80:87]
S_TMP = 88
S_LAST = 90                      # s64 .. s89 clobbered


def vreg(i):
    return "v%d" % i


def vrng(i, n):
    return "v[%d:%d]" % (i, i + n - 1)


def spair(i):
    return "s[%d:%d]" % (i, i + 1)


def X(m, i):
    return DATA0 + 8 * m + i


class I(object):
    """One emitted line (or a glued group of lines).  rd / wr: register names for the hazard tracker.  sim: closure
    executed by the simulator."""
    __slots__ = ("text", "kind", "rd", "wr", "sim", "need", "give", "nslots", "carry_rd")

    def __init__(self, text, kind, rd=(), wr=(), sim=None, need=(), give=(), carry_rd=()):
        self.text = text if isinstance(text, list) else [text]
        self.kind = kind
        self.rd = tuple(rd)
        self.wr = tuple(wr)
        self.sim = sim
        self.need = tuple(need)
        self.give = tuple(give)
        self.nslots = len(self.text)
        self.carry_rd = tuple(carry_rd)  # SGPR pairs / vcc read as a carry or lane mask by a VALU


# ---- instruction constructors (text + simulator semantics) -----------------------------------------------------------------
def v_xor_k80(d, s):
    return I("v_xor_b32_e32 v%d, s%d, v%d" % (d, S_K80, s), "valu", [vreg(s)], [vreg(d)],
             lambda st: st.setv(d, [x ^ 0x80808080 for x in st.v(s)]))


def v_mov0(d):
    return I("v_mov_b32_e32 v%d, 0" % d, "valu", [], [vreg(d)], lambda st: st.setv(d, [0] * 64))


def v_add_u32(d, a, b):
    return I("v_add_u32_e32 v%d, v%d, v%d" % (d, a, b), "valu", [vreg(a), vreg(b)], [vreg(d)],
             lambda st: st.setv(d, [(x + y) & M32 for x, y in zip(st.v(a), st.v(b))]))


def v_add_co(d, c, a, b):
    return I("v_add_co_u32_e64 v%d, %s, v%d, v%d" % (d, spair(c), a, b), "valu", [vreg(a), vreg(b)], [vreg(d), spair(c)],
             lambda st: st.addc(d, c, st.v(a), st.v(b), None))


def v_addc(d, c, a, b):
    """d = a + b + carry(c); carry out -> c.  b may be the literal 0."""
    bt = "0" if b is None else "v%d" % b
    rd = [vreg(a)] + ([] if b is None else [vreg(b)])
    return I("v_addc_co_u32_e64 v%d, %s, v%d, %s, %s" % (d, spair(c), a, bt, spair(c)), "valu", rd, [vreg(d), spair(c)],
             lambda st: st.addc(d, c, st.v(a), st.v(b) if b is not None else 0, c), carry_rd=[spair(c)])


def v_cndmask_m1(d, c):
    return I("v_cndmask_b32_e64 v%d, 0, -1, %s" % (d, spair(c)), "valu", [], [vreg(d)],
             lambda st: st.setv(d, [M32 if x else 0 for x in st.C[c]]), carry_rd=[spair(c)])


def v_and_15e(d, s):
    return I("v_and_b32_e32 v%d, 0x15e, v%d" % (d, s), "valu", [vreg(s)], [vreg(d)], lambda st: st.setv(d, [x & 0x15e for x in st.v(s)]))


def v_sub0_co(d, c, a):
    """d = 0 - a, borrow -> c"""
    return I("v_sub_co_u32_e64 v%d, %s, 0, v%d" % (d, spair(c), a), "valu", [vreg(a)], [vreg(d), spair(c)],
             lambda st: st.subb(d, c, 0, st.v(a), None))


def v_subb0(d, c, a):
    """d = a - 0 - borrow(c)"""
    return I("v_subb_co_u32_e64 v%d, %s, v%d, 0, %s" % (d, spair(c), a, spair(c)), "valu", [vreg(a)], [vreg(d), spair(c)],
             lambda st: st.subb(d, c, st.v(a), 0, c), carry_rd=[spair(c)])


def v_mul_351(d, a):
    return I("v_mul_u32_u24_e32 v%d, s%d, v%d" % (d, S_351, a), "valu", [vreg(a)], [vreg(d)],
             lambda st: st.setv(d, [(x * 351) & M32 for x in st.v(a)]))


def v_mad_u64(dp, a, sh, sreg, cp):
    """v[dp:dp+1] = v[a] * 2^sh + v[cp:cp+1]  (carry-out to vcc, unused)"""
    return I("v_mad_u64_u32 %s, vcc, v%d, s%d, %s" % (vrng(dp, 2), a, sreg, vrng(cp, 2)), "valu",
             [vreg(a), vreg(cp), vreg(cp + 1)], [vreg(dp), vreg(dp + 1), "vcc"],
             lambda st: st.mad64(dp, a, sh, cp))


def v_swap(a, b):
    return I("v_permlane32_swap_b32_e32 v%d, v%d" % (a, b), "swap", [vreg(a), vreg(b)], [vreg(a), vreg(b)],
             lambda st: st.swap(a, b))


def v_mfma(dst, fa, fb, src_c):
    """dst[16] = A(frag regs fa..fa+3) x B(regs fb..fb+3) + C (register tuple number, or the string of the asm operand)"""
    ctext = src_c if isinstance(src_c, str) else vrng(src_c, 16)
    rd = [vreg(fa + i) for i in range(4)] + [vreg(fb + i) for i in range(4)]
    if not isinstance(src_c, str):
        rd += [vreg(src_c + i) for i in range(16)]
    return I("v_mfma_i32_32x32x32_i8 %s, %s, %s, %s" % (vrng(dst, 16), vrng(fa, 4), vrng(fb, 4), ctext), "mfma", rd,
             [vreg(dst + i) for i in range(16)], lambda st: st.mfma(dst, fa, fb, src_c))


class Sched(object):
    """In-order streams, oldest-first list scheduling with the gfx950 hazard distances."""
    # minimum number of OTHER issued slots between producer and consumer
    CARRY = 2        # VALU writes SGPR pair / VCC -> VALU reads it as carry or mask
    SWAP_IN = 2      # VALU writes VGPR -> v_permlane32_swap reads it
    MFMA_IN = 2      # VALU / swap writes VGPR -> MFMA reads it as A / B
    MFMA_OUT = 13    # MFMA writes VGPR -> anything but an accumulating MFMA touches it (hipcc: s_nop 11 = 12 states)
    WAR_MFMA = 1     # MFMA reads A / B -> a later instruction overwrites them

    def __init__(self):
        self.out = []          # emitted I (incl. nops)
        self.slot = 0
        self.lastw = {}        # reg -> (slot, kind) of last write
        self.lastr_mfma = {}   # reg -> slot of last MFMA A/B read
        self.tokens = set()
        self.nops = 0
        self.rare = []         # out-of-line blocks: list of I
        self.nvmem = 0
        self.load_seq = {}
        self.last_mfma = -100

    MFMA_GAP = 5     # preferred number of other instructions between two MFMAs: back to back they hold the wave's issue for
                     # 32 cycles each, with ~6 VALU instructions in between the matrix pipe runs under them (measured:
                     # tools/ilp, "4 MFMA back-to-back + 60 mad" 10.0 vs "4 x (MFMA + 15 mad)" 9.3 cycles per instruction)

    def ready(self, ins, relaxed=False):
        if ins.kind == "mfma" and not relaxed and self.slot - self.last_mfma - 1 < self.MFMA_GAP:
            return False
        for t in ins.need:
            if t not in self.tokens:
                return False
        s = self.slot
        for r in ins.rd:
            w = self.lastw.get(r)
            if w is None:
                continue
            ws, wk = w
            gap = s - ws - 1
            if wk == "mfma":
                if ins.kind == "mfma" and r in ins.wr:
                    continue  # accumulating MFMA on its own result: no wait states
                if gap < self.MFMA_OUT:
                    return False
            elif wk == "vmem":
                pass  # covered by the explicit s_waitcnt of the consuming stream
            elif ins.kind == "swap" and wk == "valu" and gap < self.SWAP_IN:
                return False
            elif ins.kind == "mfma" and wk in ("valu", "swap") and gap < self.MFMA_IN:
                return False
        for r in ins.carry_rd:
            w = self.lastw.get(r)
            if w is not None and w[1] == "valu" and s - w[0] - 1 < self.CARRY:
                return False
        for r in ins.wr:
            w = self.lastw.get(r)
            if w is not None and w[1] == "mfma" and not (ins.kind == "mfma") and s - w[0] - 1 < self.MFMA_OUT:
                return False
            rs = self.lastr_mfma.get(r)
            if rs is not None and s - rs - 1 < self.WAR_MFMA:
                return False
        return True

    def emit(self, ins):
        if ins.kind == "vmem":
            self.nvmem += 1
            self.load_seq[ins.sim[4]] = self.nvmem          # tag -> sequence number of its latest load
        if ins.kind == "wait":
            # loads complete in order: everything up to the last load of this butterfly's fragments must have landed, the
            # younger ones (the next butterfly's prefetch) may stay in flight
            n = self.nvmem - self.load_seq[ins.sim[1]]
            ins.text = ["s_waitcnt vmcnt(%d)" % n]
            ins.sim = ("waitcnt", n)
        self.out.append(ins)
        for r in ins.wr:
            self.lastw[r] = (self.slot + ins.nslots - 1, ins.kind)
        if ins.kind == "mfma":
            self.last_mfma = self.slot
            for r in ins.rd[:8]:
                self.lastr_mfma[r] = self.slot
        self.slot += ins.nslots
        for t in ins.give:
            self.tokens.add(t)

    def nop(self):
        self.out.append(I("s_nop 0", "nop"))
        self.slot += 1
        self.nops += 1

    def run(self, streams):
        """streams: list of (priority key, [I...]); consumed in place."""
        streams = sorted(streams, key=lambda kv: kv[0])
        heads = [0] * len(streams)
        left = sum(len(s[1]) for s in streams)
        stall = 0
        while left:
            done = False
            for relaxed in (False, True):   # an MFMA closer than MFMA_GAP to the previous one only instead of an s_nop
                for k, (_, lst) in enumerate(streams):
                    h = heads[k]
                    if h < len(lst) and self.ready(lst[h], relaxed):
                        self.emit(lst[h])
                        heads[k] = h + 1
                        left -= 1
                        stall = 0
                        done = True
                        break
                if done:
                    break
            if not done:
                self.nop()
                stall += 1
                if stall > 64:
                    pend = [(streams[k][0], streams[k][1][heads[k]].text, streams[k][1][heads[k]].need)
                            for k in range(len(streams)) if heads[k] < len(streams[k][1])]
                    raise RuntimeError("scheduler deadlock: %r" % (pend[:6],))


# ---- streams of one butterfly ------------------------------------------------------------------------------------------
class Label(object):
    count = 0

    @classmethod
    def new(cls, stem):
        cls.count += 1
        return "%s_%d_" % (stem, cls.count)


def rare_check(sched, pair, body, tag):
    """s_cmp + branch to an out-of-line block that runs `body` (list of I, executed strictly in order with wait states
    inserted) and comes back.  Returns the glued in-line instruction."""
    lab, back = Label.new("RARE" + tag), Label.new("BACK" + tag)
    blk = [I("%s%%=:" % lab, "label")]
    for k, ins in enumerate(body):
        blk.append(I("s_nop 1", "nop"))
        blk.append(ins)
    blk.append(I("s_nop 1", "nop"))
    blk.append(I("s_branch %s%%=" % back, "branch", sim=("jump", back)))
    sched.rare.append(blk)
    return I(["s_cmp_lg_u64 %s, 0" % spair(pair), "s_cbranch_scc1 %s%%=" % lab, "%s%%=:" % back], "salu", [spair(pair)], ["scc"],
             ("rare", pair, lab, back))


def stream_P(sched, j, m0, m1, need, give_a):
    """TA = a ^ K, a += b (fold), b ^= K, half swaps.  a = x[m0], b = x[m1]."""
    a = [X(m0, i) for i in range(8)]
    b = [X(m1, i) for i in range(8)]
    L = []
    xa = [v_xor_k80(TA + i, a[i]) for i in range(8)]
    xb = [v_xor_k80(b[i], b[i]) for i in range(8)]
    s = [v_add_co(a[0], S_CA, a[0], b[0])] + [v_addc(a[i], S_CA, a[i], b[i]) for i in range(1, 8)]
    # order: xa_i before s_i, xb_i after s_i; two fillers between consecutive carry instructions
    L += [xa[0], xa[1], s[0], xa[2], xa[3], s[1], xa[4], xa[5], s[2], xa[6], xa[7], s[3], xb[0], xb[1], s[4], xb[2], xb[3],
          s[5], xb[4], xb[5], s[6], xb[6], s[7], xb[7]]
    L[0].need = tuple(need)
    # fold of the carry out of limb 7: + c on limbs 0..1 (c = 2^256 - p = 0x15e_ffffffff)
    L.append(v_cndmask_m1(SM, S_CA))
    L.append(v_add_co(a[0], S_CA, a[0], SM))
    L.append(v_and_15e(SM, SM))
    L.append(v_addc(a[1], S_CA, a[1], SM))
    # rare: carry out of limb 1 (limb 1 >= 2^32 - 351): propagate; a second wrap leaves a value < 2^42 -> + c again
    body = [v_addc(a[i], S_CA, a[i], None) for i in range(2, 8)]
    body += [v_cndmask_m1(SM, S_CA), v_add_co(a[0], S_CA, a[0], SM), v_and_15e(SM, SM), v_addc(a[1], S_CA, a[1], SM),
             v_addc(a[2], S_CA, a[2], None)]
    chk = rare_check(sched, S_CA, body, "S")
    chk.give = tuple(give_a) + ("SUM%d" % j,)
    L.append(chk)
    # swaps: TA[i] <-> TA[4+i] and b[i] <-> b[4+i] (upper 32 lanes of the first with lower 32 lanes of the second)
    for i in range(4):
        L.append(v_swap(TA + i, TA + 4 + i))
    for i in range(4):
        L.append(v_swap(b[i], b[4 + i]))
    L[-1].give = ("P%d" % j,)
    return L


def norm_chain(acc, ez, t):
    """16 non-negative sums at byte spacing (accumulator registers acc..acc+15) -> limbs in acc+0, +4, +8, +12 and the
    carry out in acc+13.  Per limb m: e = s0 + carry-in, then three v_mad_u64_u32 add s1 << 8, s2 << 16, s3 << 24; the last
    one writes the (even-aligned) pair (acc+4m, acc+4m+1) = (limb, carry into the next limb) over the consumed sums."""
    L = []
    for m in range(4):
        s0, s1, s2, s3 = acc + 4 * m, acc + 4 * m + 1, acc + 4 * m + 2, acc + 4 * m + 3
        cin = ez + 1 if m == 0 else acc + 4 * (m - 1) + 1    # v[ez+1] == 0
        L.append(v_add_u32(ez, s0, cin))                     # < 2^32: sums < 2^22, carry < 2^14
        L.append(v_mad_u64(t, s1, 8, S_SH8, ez))
        L.append(v_mad_u64(t, s2, 16, S_SH16, t))
        L.append(v_mad_u64(acc + 4 * m, s3, 24, S_SH24, t))
    return L


def stream_N(j):
    c1 = norm_chain(ACC1, EZ1, T1)
    c2 = norm_chain(ACC2, EZ2, T2)
    L = []
    for x, y in zip(c1, c2):
        L += [x, y]
    L[-1].give = ("N%d" % j,)
    return L


def stream_M(j, m1, wait, need, same_frag=False, fr=FRAG):
    """the four MFMAs: acc1 = W1 x a1 + OFFS, acc2 = W2 x a2 + OFFS, acc1 += N1 x b1, acc2 += N2 x b2
    (same_frag: both column groups use (W1, N1) -- stage 2, where the twiddle does not depend on the lane half)"""
    L = []
    if wait:
        L.append(I("s_waitcnt vmcnt(?)", "wait", [], [], ("waitcnt?", j)))
    w2, n2 = (fr + 0, fr + 4) if same_frag else (fr + 8, fr + 12)
    L.append(v_mfma(ACC1, fr + 0, TA + 0, "%[offs]"))
    L.append(v_mfma(ACC2, w2, TA + 4, "%[offs]"))
    L[-1].give = ("M12_%d" % j,)
    L.append(v_mfma(ACC1, fr + 4, X(m1, 0), ACC1))
    L.append(v_mfma(ACC2, n2, X(m1, 4), ACC2))
    L[-1].give = ("M%d" % j,)
    L[0].need = tuple(need)
    return L


def stream_J(sched, j, m1, give_b):
    """swaps -> acc1[0,4,8,12 | 13] = own low half | carry, acc2[...] = own high half | carry; b = low + high << 128 with the
    top carry T folded: T 2^256 == T c = (351 T) << 32 - T."""
    b = [X(m1, i) for i in range(8)]
    lo = [ACC1 + 0, ACC1 + 4, ACC1 + 8, ACC1 + 12]
    hi = [ACC2 + 0, ACC2 + 4, ACC2 + 8, ACC2 + 12]
    clo, chi = ACC1 + 13, ACC2 + 13
    L = [v_swap(clo, chi)] + [v_swap(lo[i], hi[i]) for i in range(4)]
    L.append(v_add_co(b[4], S_CJ, hi[0], clo))
    for i in range(1, 4):
        L.append(v_addc(b[4 + i], S_CJ, hi[i], None))
    L.append(v_addc(JT, S_CJ, chi, None))             # T < 2^15 (no carry out)
    L.append(v_mul_351(JD1, JT))
    L.append(v_sub0_co(JD0, S_CJ, JT))
    L.append(v_subb0(JD1, S_CJ, JD1))
    L.append(v_add_co(b[0], S_CJ, lo[0], JD0))
    L.append(v_addc(b[1], S_CJ, lo[1], JD1))
    L.append(v_addc(b[2], S_CJ, lo[2], None))
    L.append(v_addc(b[3], S_CJ, lo[3], None))
    L[-1].give = ("ACC%d" % j,)                       # last read of the accumulators
    # rare: carry out of limb 3 (needs limbs 2, 3 all ones: ~2^-64 per lane; taken by crafted vectors only)
    body = [v_addc(b[i], S_CJ, b[i], None) for i in range(4, 8)]
    body += [v_cndmask_m1(JM, S_CJ), v_add_co(b[0], S_CJ, b[0], JM), v_and_15e(JM, JM), v_addc(b[1], S_CJ, b[1], JM),
             v_addc(b[2], S_CJ, b[2], None)]
    chk = rare_check(sched, S_CJ, body, "J")
    chk.give = tuple(give_b) + ("J%d" % j,)
    L.append(chk)
    return L


def stream_F(j, base_sreg, off1, off2, need, both, fr=FRAG):
    """TwMat addresses and the fragment loads of butterfly j: AD1 = base + off1 (column group 1), AD2 = base + off2."""
    L = []

    def addr(dst, off):
        if isinstance(off, str):   # byte offset in an asm operand (an SGPR): register groups of the LDS-resident tile
            return I(["s_add_u32 s%d, s%d, %%[%s]" % (dst, base_sreg, off), "s_addc_u32 s%d, s%d, 0" % (dst + 1, base_sreg + 1)],
                     "salu", [], ["scc", "s%d" % dst], ("saddr_op", dst, base_sreg, off))
        return I(["s_add_u32 s%d, s%d, 0x%x" % (dst, base_sreg, off), "s_addc_u32 s%d, s%d, 0" % (dst + 1, base_sreg + 1)], "salu",
                 [], ["scc", "s%d" % dst], ("saddr", dst, base_sreg, off))

    def load(dreg, areg, imm):
        return I("global_load_dwordx4 %s, %%[lane16], %s offset:%d" % (vrng(dreg, 4), spair(areg), imm), "vmem",
                 ["s%d" % areg], [vreg(dreg + i) for i in range(4)], ("load", dreg, areg, imm, j))

    L.append(addr(S_AD1, off1))
    L[0].need = tuple(need)
    if both:
        L.append(addr(S_AD2, off2))
    L.append(load(fr + 0, S_AD1, 0))
    L.append(load(fr + 4, S_AD1, 1024))
    if both:
        L.append(load(fr + 8, S_AD2, 0))
        L.append(load(fr + 12, S_AD2, 1024))
    return L


def stream_AS(sched, j, m0, m1, need, give):
    """twiddle 1 (stage 2): (a, b) -> (a + b, a - b), two carry chains side by side, both lazily reduced.  The difference is
    built in the idle fragment registers and copied over b (b feeds both chains)."""
    a = [X(m0, i) for i in range(8)]
    b = [X(m1, i) for i in range(8)]
    d = [FRAG_B + i for i in range(8)]     # stage 2 needs 8 fragment registers per twiddle: its second buffer is FRAG + 8, this is free
    L = []
    sub = [I("v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (d[0], spair(S_CB), a[0], b[0]), "valu", [vreg(a[0]), vreg(b[0])],
             [vreg(d[0]), spair(S_CB)], (lambda dd, aa, bb: (lambda st: st.subb(dd, S_CB, st.v(aa), st.v(bb), None)))(d[0], a[0], b[0]))]
    for i in range(1, 8):
        sub.append(I("v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (d[i], spair(S_CB), a[i], b[i], spair(S_CB)), "valu",
                     [vreg(a[i]), vreg(b[i])], [vreg(d[i]), spair(S_CB)],
                     (lambda dd, aa, bb: (lambda st: st.subb(dd, S_CB, st.v(aa), st.v(bb), S_CB)))(d[i], a[i], b[i]),
                     carry_rd=[spair(S_CB)]))
    add = [v_add_co(a[0], S_CA, a[0], b[0])] + [v_addc(a[i], S_CA, a[i], b[i]) for i in range(1, 8)]
    for i in range(8):           # sub_i reads a_i before add_i overwrites it
        L += [sub[i], add[i]]
    L[0].need = tuple(need)
    # folds: sum + c on a carry, difference - c on a borrow
    L.append(v_cndmask_m1(SM, S_CA))
    L.append(v_cndmask_m1(AM, S_CB))
    L.append(v_add_co(a[0], S_CA, a[0], SM))
    L.append(I("v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (d[0], spair(S_CB), d[0], AM), "valu", [vreg(d[0]), vreg(AM)],
               [vreg(d[0]), spair(S_CB)], lambda st: st.subb(d[0], S_CB, st.v(d[0]), st.v(AM), None)))
    L.append(v_and_15e(SM, SM))
    L.append(v_and_15e(AM, AM))
    L.append(v_addc(a[1], S_CA, a[1], SM))
    L.append(I("v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (d[1], spair(S_CB), d[1], AM, spair(S_CB)), "valu",
               [vreg(d[1]), vreg(AM)], [vreg(d[1]), spair(S_CB)], lambda st: st.subb(d[1], S_CB, st.v(d[1]), st.v(AM), S_CB),
               carry_rd=[spair(S_CB)]))
    body = [v_addc(a[i], S_CA, a[i], None) for i in range(2, 8)]
    body += [v_cndmask_m1(SM, S_CA), v_add_co(a[0], S_CA, a[0], SM), v_and_15e(SM, SM), v_addc(a[1], S_CA, a[1], SM),
             v_addc(a[2], S_CA, a[2], None)]
    L.append(rare_check(sched, S_CA, body, "A"))

    def subb0(r):
        return I("v_subb_co_u32_e64 v%d, %s, v%d, 0, %s" % (r, spair(S_CB), r, spair(S_CB)), "valu", [vreg(r)],
                 [vreg(r), spair(S_CB)], lambda st: st.subb(r, S_CB, st.v(r), 0, S_CB), carry_rd=[spair(S_CB)])

    # rare borrow out of limb 1: propagate; a second borrow means the value was < c: subtract c once more (limbs 0..2)
    body = [subb0(d[i]) for i in range(2, 8)]
    body += [v_cndmask_m1(AM, S_CB),
             I("v_sub_co_u32_e64 v%d, %s, v%d, v%d" % (d[0], spair(S_CB), d[0], AM), "valu", [vreg(d[0]), vreg(AM)],
               [vreg(d[0]), spair(S_CB)], lambda st: st.subb(d[0], S_CB, st.v(d[0]), st.v(AM), None)),
             v_and_15e(AM, AM),
             I("v_subb_co_u32_e64 v%d, %s, v%d, v%d, %s" % (d[1], spair(S_CB), d[1], AM, spair(S_CB)), "valu",
               [vreg(d[1]), vreg(AM)], [vreg(d[1]), spair(S_CB)], lambda st: st.subb(d[1], S_CB, st.v(d[1]), st.v(AM), S_CB),
               carry_rd=[spair(S_CB)])]
    L.append(rare_check(sched, S_CB, body, "B"))
    for i in range(8):
        L.append(I("v_mov_b32_e32 v%d, v%d" % (b[i], d[i]), "valu", [vreg(d[i])], [vreg(b[i])],
                   (lambda dd, ss: (lambda st: st.setv(dd, st.v(ss))))(b[i], d[i])))
    L[-1].give = tuple(give) + ("SUM%d" % j,)
    return L


# ---- stages ---------------------------------------------------------------------------------------------------------------
def prologue(stage, log_r):
    L = [I("s_mov_b32 s%d, 0x80808080" % S_K80, "salu"), I("s_movk_i32 s%d, 0x100" % S_SH8, "salu"),
         I("s_mov_b32 s%d, 0x10000" % S_SH16, "salu"), I("s_mov_b32 s%d, 0x1000000" % S_SH24, "salu"),
         I("s_movk_i32 s%d, 0x15f" % S_351, "salu"), v_mov0(EZ1 + 1), v_mov0(EZ2 + 1)]
    if stage == 1:
        for k, mu in enumerate((3, 2, 1, 0)):   # base_mu = mats + (rho_lo << (3 - mu)) * sizeof(TwMat)
            L.append(I(["s_lshl_b32 s%d, %%[rho], %d" % (S_TMP, 14 - mu),
                        "s_add_u32 s%d, %%[mlo], s%d" % (S_BASE + 2 * k, S_TMP),
                        "s_addc_u32 s%d, %%[mhi], 0" % (S_BASE + 2 * k + 1)], "salu", [], ["scc"], ("base", k, 14 - mu)))
    else:
        L.append(I(["s_mov_b32 s%d, %%[mlo]" % S_BASE, "s_mov_b32 s%d, %%[mhi]" % (S_BASE + 1)], "salu", [], [], ("base0",)))
    return L


def butterflies(stage, log_r):
    """[(m0, m1, off1, off2 or None, base sreg)], off = byte offset of the TwMat(s); None twiddle = 1 (stage 2 only)."""
    R = 1 << log_r
    out = []
    if stage == 1:
        G = R // 16
        for k, mu in enumerate((3, 2, 1, 0)):
            m0s = [((b >> mu) << (mu + 1)) | (b & ((1 << mu) - 1)) for b in range(8)]
            m0s.sort(key=lambda m: (m & ((1 << mu) - 1), m))
            for m0 in m0s:
                eb = (m0 & ((1 << mu) - 1)) * G
                off1 = (eb << (3 - mu)) * 2048
                out.append((m0, m0 | (1 << mu), off1, off1 + (1 << (3 - mu)) * 2048, S_BASE + 2 * k))
    else:
        qmax = min(log_r - 5, 3)
        for q in range(qmax, -1, -1):
            m0s = [((b >> q) << (q + 1)) | (b & ((1 << q) - 1)) for b in range(8)]
            m0s.sort(key=lambda m: (-(m & ((1 << q) - 1)), m))   # twiddle 1 (plain add / sub) last: it fills the tail
            for m0 in m0s:
                e = (m0 & ((1 << q) - 1)) << (log_r - 1 - q)
                out.append((m0, m0 | (1 << q), e * 2048 if e else None, None, S_BASE))
    return out


def build_stage(stage, log_r, bf=None):
    Label.count = 0
    sched = Sched()
    for ins in prologue(stage, log_r):
        sched.emit(ins)
    if bf is None:
        bf = butterflies(stage, log_r)
    ver = [0] * 16            # version of x[m] (token "X<m>v<k>" = version k is final)
    for m in range(16):
        sched.tokens.add("X%dv0" % m)
    streams = []
    prev_mf = None            # index of the previous MFMA butterfly
    prev_tw = None
    nload = 0                 # fragment sets loaded so far: set k lives in buffer k & 1
    last_user = [None, None]  # last MFMA butterfly that read each buffer
    fr = FRAG
    for j, (m0, m1, off1, off2, base) in enumerate(bf):
        need_x = ["X%dv%d" % (m0, ver[m0]), "X%dv%d" % (m1, ver[m1])]
        if j:
            need_x.append("SUM%d" % (j - 1))  # the sum chains share a carry pair and a mask register: one at a time
        ver[m0] += 1
        ver[m1] += 1
        give_a, give_b = ["X%dv%d" % (m0, ver[m0])], ["X%dv%d" % (m1, ver[m1])]
        if off1 is None:
            streams.append(((j, 0), stream_AS(sched, j, m0, m1, need_x, give_a + give_b)))
            continue
        tw = (base, off1, off2)
        load = tw != prev_tw
        prev_tw = tw
        if load:
            buf = nload & 1
            nload += 1
            # stage 1: two sets of 16 (W1 N1 W2 N2); stage 2 (one twiddle for both column groups): two sets of 8 in FRAG
            fr = (FRAG, FRAG_B)[buf] if stage == 1 else FRAG + 8 * buf
        # P(j) writes TA: after the previous MFMA butterfly's first two MFMAs; J(j)/N(j) use the accumulators
        streams.append(((j, 1), stream_P(sched, j, m0, m1, need_x + (["M12_%d" % prev_mf] if prev_mf is not None else []), give_a)))
        if load:
            # the buffer is free once the last butterfly that read it has issued its MFMAs; priority (j - 1): the loads go out
            # a whole butterfly before their use
            streams.append(((j - 1, -1), stream_F(j, base, off1, off2 if off2 is not None else 0,
                                                  ["M%d" % last_user[buf]] if last_user[buf] is not None else [],
                                                  off2 is not None, fr)))
        last_user[buf] = j
        streams.append(((j, 2), stream_M(j, m1, load, ["P%d" % j] + (["ACC%d" % prev_mf] if prev_mf is not None else []),
                                         same_frag=off2 is None, fr=fr)))
        n = stream_N(j)
        n[0].need = ("M%d" % j,)
        streams.append(((j, 3), n))
        jj = stream_J(sched, j, m1, give_b)
        jj[0].need = ("N%d" % j,)
        streams.append(((j, 4), jj))
        prev_mf = j
    sched.run(streams)
    # the block ends clean: nothing the compiler's next instruction could trip over
    sched.emit(I("s_nop 1", "nop"))
    return sched, bf


# ---- simulator ------------------------------------------------------------------------------------------------------------
def signed_digits(v):
    """32 digits in [-128, 127] with sum d_m 256^m == v (mod p) for canonical v (mfma_tw.cuh:signed_digits)."""
    u = list(v.to_bytes(32, "little")) + [0]
    small = True
    for m in range(31, -1, -1):
        if u[m] != 0x7f:
            small = u[m] < 0x7f
            break
    if not small:
        t = v + (2**256 - P)
        assert t < 2**256
        u = list(t.to_bytes(32, "little")) + [0xff]
    d, carry = [], 0
    for m in range(32):
        t = u[m] + carry
        if t >= 128:
            d.append(t - 256)
            carry = 1
        else:
            d.append(t)
            carry = 0
    top = u[32] - 256 if u[32] >= 128 else u[32]
    assert top + carry == 0
    assert (sum(x << (8 * m) for m, x in enumerate(d)) - v) % P == 0
    return d


def twmat_bytes(w):
    """TwMat image (2048 bytes: w[64][16], nw[64][16]) of the twiddle w, as shk_build_twmat lays it out."""
    rho = lambda i: 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3)
    out = bytearray(2048)
    for part, val in ((0, w % P), (1, (-w) % P)):
        t = val
        dig = []
        for kappa in range(32):
            dig.append(signed_digits(t))
            t = t * 256 % P
        for lane in range(64):
            i, h = lane & 31, lane >> 5
            for jj in range(16):
                out[part * 1024 + lane * 16 + jj] = dig[16 * h + jj][rho(i)] & 0xff
    return bytes(out)


def offs_block():
    """[half][r] = O_(16 half + r) = 2^20 + delta: every row sum becomes non-negative, sum O_j 2^(8j) == 0 (mod p)."""
    delta = (-(2**20) * ((2**256 - 1) // 255)) % P
    db = delta.to_bytes(32, "little")
    O = [(1 << 20) + db[j] for j in range(32)]
    assert sum(o << (8 * j) for j, o in enumerate(O)) % P == 0
    return [[O[16 * h + r] for r in range(16)] for h in range(2)]


class Sim(object):
    def __init__(self, mem, mats_addr, rho_lo):
        self.V = {}
        self.C = {}
        self.S = {}
        self.mem = mem
        self.mats_addr = mats_addr
        self.rho = rho_lo
        O = offs_block()
        self.offs = [[O[l >> 5][r] for l in range(64)] for r in range(16)]
        self.inflight = []
        self.operands = {}

    def v(self, r):
        return self.V[r]

    def setv(self, r, val):
        self.V[r] = list(val)

    def addc(self, d, c, a, b, cin):
        a = [a] * 64 if isinstance(a, int) else a
        b = [b] * 64 if isinstance(b, int) else b
        ci = self.C[cin] if cin is not None else [0] * 64
        res, co = [], []
        for l in range(64):
            t = a[l] + b[l] + (1 if ci[l] else 0)
            res.append(t & M32)
            co.append(t >> 32)
        self.V[d] = res
        self.C[c] = co

    def subb(self, d, c, a, b, bin_):
        a = [a] * 64 if isinstance(a, int) else a
        b = [b] * 64 if isinstance(b, int) else b
        bi = self.C[bin_] if bin_ is not None else [0] * 64
        res, bo = [], []
        for l in range(64):
            t = a[l] - b[l] - (1 if bi[l] else 0)
            res.append(t & M32)
            bo.append(1 if t < 0 else 0)
        self.V[d] = res
        self.C[c] = bo

    def mad64(self, dp, a, sh, cp):
        lo, hi = [], []
        A, CL, CH = self.V[a], self.V[cp], self.V[cp + 1]
        for l in range(64):
            t = (A[l] << sh) + CL[l] + (CH[l] << 32)
            assert t < 2**64
            lo.append(t & M32)
            hi.append(t >> 32)
        self.V[dp], self.V[dp + 1] = lo, hi

    def swap(self, a, b):
        A, B = list(self.V[a]), list(self.V[b])
        for l in range(32):
            A[32 + l], B[l] = B[l], A[32 + l]
        self.V[a], self.V[b] = A, B

    def mfma(self, dst, fa, fb, src_c):
        def s8(x):
            return x - 256 if x >= 128 else x
        Amat = [[0] * 32 for _ in range(32)]   # [i][k]
        Bmat = [[0] * 32 for _ in range(32)]   # [k][n]
        for lane in range(64):
            i, h = lane & 31, lane >> 5
            for q in range(4):
                wa, wb = self.V[fa + q][lane], self.V[fb + q][lane]
                for bb in range(4):
                    k = 16 * h + 4 * q + bb
                    Amat[i][k] = s8((wa >> (8 * bb)) & 0xff)
                    Bmat[k][i] = s8((wb >> (8 * bb)) & 0xff)
        res = [[0] * 64 for _ in range(16)]
        for lane in range(64):
            n, h = lane & 31, lane >> 5
            for r in range(16):
                i = (r & 3) + 8 * (r >> 2) + 4 * h
                acc = sum(Amat[i][k] * Bmat[k][n] for k in range(32))
                if isinstance(src_c, str):
                    acc += self.offs[r][lane]
                else:
                    cv = self.V[src_c + r][lane]
                    acc += cv - (1 << 32) if cv >= (1 << 31) else cv
                assert -(1 << 31) <= acc < (1 << 31)
                res[r][lane] = acc & M32
        for r in range(16):
            self.V[dst + r] = res[r]

    def run(self, sched):
        prog = []
        labels = {}
        for blk in [sched.out] + sched.rare:
            for ins in blk:
                if ins.kind == "label":
                    labels[ins.text[0].split("%")[0]] = len(prog)
                prog.append(ins)
            prog.append(None)  # end of main / fallthrough guard
        pc, steps = 0, 0
        while True:
            ins = prog[pc]
            if ins is None:
                if pc == len(sched.out):
                    return
                raise RuntimeError("fell off a rare block")
            pc += 1
            steps += 1
            sm = ins.sim
            if sm is None:
                continue
            if callable(sm):
                sm(self)
            elif sm[0] == "rare":
                _, pair, lab, back = sm
                labels[back] = pc
                if any(self.C[pair]):
                    self.rare_taken = getattr(self, "rare_taken", 0) + 1
                    pc = labels[lab]
            elif sm[0] == "jump":
                pc = labels[sm[1]]
            elif sm[0] == "base":
                _, k, sh = sm
                self.S[S_BASE + 2 * k] = self.mats_addr + (self.rho << sh)
            elif sm[0] == "base0":
                self.S[S_BASE] = self.mats_addr
            elif sm[0] == "saddr":
                _, dst, base, off = sm
                self.S[dst] = self.S[base] + off
            elif sm[0] == "saddr_op":
                _, dst, base, opname = sm
                self.S[dst] = self.S[base] + self.operands[opname]
            elif sm[0] == "load":
                _, dreg, areg, imm, _tag = sm
                vals = {}
                for q in range(4):
                    vals[dreg + q] = [int.from_bytes(self.mem[self.S[areg] + imm + 16 * l + 4 * q:
                                                              self.S[areg] + imm + 16 * l + 4 * q + 4], "little")
                                      for l in range(64)]
                    self.V[dreg + q] = [0xdeadbeef] * 64   # in flight: whoever reads it before the wait gets garbage
                self.inflight.append(vals)
            elif sm[0] == "waitcnt":
                keep = sm[1]
                while len(self.inflight) > keep:
                    for r, val in self.inflight.pop(0).items():
                        self.V[r] = val
            else:
                raise RuntimeError(sm)


def selftest(stage, log_r, seed=1, crafted=False, verbose=False):
    rng = random.Random(seed * 1000 + stage * 10 + log_r)
    R = 1 << log_r
    sched, bf = build_stage(stage, log_r)
    wR = pow(7, (P - 1) // R, P)
    tw = [pow(wR, i, P) for i in range(R // 2)]
    mats_addr = 0x10000
    mem = bytearray(mats_addr) + b"".join(twmat_bytes(t) for t in tw)
    waves = max(1, R // 32)
    wave = rng.randrange(waves)
    st = Sim(mem, mats_addr, 2 * wave)
    st.S_lane16 = None
    # lane16 operand: the simulator adds 16 * lane inside "load"
    vals = [[rng.randrange(2**256) for _ in range(64)] for _ in range(16)]
    if crafted:
        for m in range(16):
            for l in range(64):
                c = rng.randrange(6)
                if c == 0:
                    vals[m][l] = 2**256 - 1 - rng.randrange(4)
                elif c == 1:
                    vals[m][l] = rng.randrange(4)
                elif c == 2:
                    vals[m][l] = P - 1 - rng.randrange(3)
                elif c == 3:
                    vals[m][l] = (2**256 - rng.randrange(1 << 40)) % 2**256
    for m in range(16):
        for i in range(8):
            st.V[X(m, i)] = [(vals[m][l] >> (32 * i)) & M32 for l in range(64)]
    st.run(sched)
    # reference: the same butterflies on integers
    ref = [list(v) for v in vals]
    G = R // 16
    for (m0, m1, off1, off2, base) in bf:
        for l in range(64):
            hb = l >> 5
            if off1 is None:
                w = 1
            elif stage == 1:
                k = 3 - (base - S_BASE) // 2
                idx = (off1 // 2048) + ((2 * wave + hb) << (3 - (3 - (base - S_BASE) // 2)))
                w = tw[idx]
            else:
                w = tw[off1 // 2048]
            a, b = ref[m0][l], ref[m1][l]
            ref[m0][l] = (a + b) % P
            ref[m1][l] = (a - b) * w % P
    bad = 0
    for m in range(16):
        for l in range(64):
            got = sum(st.V[X(m, i)][l] << (32 * i) for i in range(8))
            if got % P != ref[m][l]:
                bad += 1
    ninstr = sum(i.nslots for i in sched.out if i.kind not in ("nop", "label"))
    if verbose:
        print("stage %d R=2^%d: %d slots, %d s_nop, %d butterflies, rare blocks taken %d, mismatches %d" %
              (stage, log_r, ninstr, sched.nops, len(bf), getattr(st, "rare_taken", 0), bad))
    return bad, sched


def asm_text(sched):
    lines = []
    for ins in sched.out:
        lines += ins.text
    lines.append("s_branch SHKEND%=")
    for blk in sched.rare:
        for ins in blk:
            lines += ins.text
    lines.append("SHKEND%=:")
    return lines


def emit_inc(path):
    O = offs_block()
    L = ["// GENERATED by gen_bflyasm.py -- do not edit.  The butterfly stages of ntt_mfma.hip as scheduled gfx950 asm blocks.",
         "// Registers: x[m] is pinned to v[%d + 8 m : %d + 8 m]; v0..v%d, s%d..s%d, vcc and scc are clobbered." %
         (DATA0, DATA0 + 7, NTEMP - 1, S_CA, S_LAST - 1),
         "typedef uint32_t shk_x8 __attribute__((ext_vector_type(8)));",
         "// accumulator offsets O_(16 half + r) = 2^20 + delta (sum O_j 2^(8j) == 0 mod p): srcC of the first MFMA of a group",
         "__device__ static const int32_t SHK_OFFS[2][16] = {{%s}, {%s}};" %
         (", ".join(str(v) for v in O[0]), ", ".join(str(v) for v in O[1]))]
    clob = ['"v%d"' % i for i in range(NTEMP)] + ['"s%d"' % i for i in range(S_CA, S_LAST)] + ['"vcc"', '"scc"']
    L.append("#define SHK_BFLY_CLOBBERS " + ", ".join(clob))
    stats = []
    for stage in (1, 2):
        for log_r in (5, 6, 7, 8):
            sched, bf = build_stage(stage, log_r)
            n = sum(i.nslots for i in sched.out if i.kind not in ("nop", "label"))
            stats.append((stage, log_r, n, sched.nops, len(bf)))
            L.append("// stage %d of a radix-2^%d tile: %d butterflies, %d instruction slots, %d s_nop" % (stage, log_r, len(bf), n, sched.nops))
            args = "shk_x8 (&x)[16], const shk_v16i& offs, uint32_t lane16, uint32_t mlo, uint32_t mhi" + (", uint32_t rho" if stage == 1 else "")
            L.append("__device__ __forceinline__ void shk_stage%d_asm_%d(%s) {" % (stage, log_r, args))
            L.append("  asm volatile(")
            for t in asm_text(sched):
                L.append('      "%s\\n\\t"' % t)
            L.append("      : " + ", ".join('"+{v[%d:%d]}"(x[%d])' % (DATA0 + 8 * m, DATA0 + 8 * m + 7, m) for m in range(16)))
            ins = '[offs] "v"(offs), [lane16] "v"(lane16), [mlo] "s"(mlo), [mhi] "s"(mhi)' + (', [rho] "s"(rho)' if stage == 1 else "")
            L.append("      : " + ins)
            L.append("      : SHK_BFLY_CLOBBERS);")
            L.append("}")
    with open(path, "w") as fh:
        fh.write("\n".join(L) + "\n")
    return stats


# ---- register groups of an LDS-resident tile (ntt_kernels.cuh with MFMA butterflies) -------------------------------------------
# Four elements per thread (x[h] pinned to v[96 + 8 h ..]), two DIF levels = up to four butterflies.  The twiddle of a butterfly
# is the same for the 32 lanes of a half-wave (they are the 32 columns of one row); the kernel passes the byte offsets of the two
# TwMat images (lanes 0..31 / 32..63) of every MFMA butterfly as scalar operands o<k>lo / o<k>hi.
GROUP_PRIO = int(os.environ.get("SHK_GROUP_PRIO", "0"))  # experiment: s_setprio around a block (measured: no effect)
GROUP_PATTERNS = {
    # name: [(m0, m1, kind)], kind "M" = MFMA butterfly with a table twiddle, "A" = twiddle 1 (add / sub)
    "MMMM": [(0, 2, "M"), (1, 3, "M"), (0, 1, "M"), (2, 3, "M")],   # both levels general
    "AMAA": [(0, 2, "A"), (1, 3, "M"), (0, 1, "A"), (2, 3, "A")],   # the last two levels of a tile: 1, w^(R/4), 1, 1
    "AA": [(0, 1, "A"), (2, 3, "A")],                               # a single last level (odd log R)
    "MM": [(0, 1, "M"), (2, 3, "M")],                               # a single general level
}


def with_group_layout(fn):
    def wrapped(*a, **k):
        global DATA0, FRAG_B, NTEMP
        saved = (DATA0, FRAG_B, NTEMP)
        DATA0, FRAG_B, NTEMP = 96, FRAG, 70
        try:
            return fn(*a, **k)
        finally:
            DATA0, FRAG_B, NTEMP = saved
    return wrapped


@with_group_layout
def build_group(pattern):
    """the scheduled block of one register group; returns (sched, butterfly list [(m0, m1, slot or None)])"""
    Label.count = 0
    sched = Sched()
    if GROUP_PRIO:
        sched.emit(I("s_setprio %d" % GROUP_PRIO, "salu"))   # a latency-bound block between VALU-dense waves: issue first
    for ins in prologue(2, 0):   # constants + base = the TwMat table
        sched.emit(ins)
    bf = []
    slot = 0
    for (m0, m1, kind) in GROUP_PATTERNS[pattern]:
        if kind == "M":
            bf.append((m0, m1, "o%dlo" % slot, "o%dhi" % slot, S_BASE))
            slot += 1
        else:
            bf.append((m0, m1, None, None, S_BASE))
    ver = [0] * 16
    for m in range(16):
        sched.tokens.add("X%dv0" % m)
    streams = []
    prev_mf = prev_as = None
    for j, (m0, m1, off1, off2, base) in enumerate(bf):
        need_x = ["X%dv%d" % (m0, ver[m0]), "X%dv%d" % (m1, ver[m1])]
        if j:
            need_x.append("SUM%d" % (j - 1))
        ver[m0] += 1
        ver[m1] += 1
        give_a, give_b = ["X%dv%d" % (m0, ver[m0])], ["X%dv%d" % (m1, ver[m1])]
        if off1 is None:
            # the difference is built in the fragment registers (no second buffer in this layout): not while an MFMA
            # butterfly's fragments live there
            streams.append(((j, 0), stream_AS(sched, j, m0, m1, need_x + (["M%d" % prev_mf] if prev_mf is not None else []),
                                              give_a + give_b)))
            prev_as = j
            continue
        streams.append(((j, 1), stream_P(sched, j, m0, m1, need_x + (["M12_%d" % prev_mf] if prev_mf is not None else []), give_a)))
        streams.append(((j, -1), stream_F(j, base, off1, off2, (["M%d" % prev_mf] if prev_mf is not None else []) +
                                          (["SUM%d" % prev_as] if prev_as is not None else []), True, FRAG)))
        streams.append(((j, 2), stream_M(j, m1, True, ["P%d" % j] + (["ACC%d" % prev_mf] if prev_mf is not None else []), fr=FRAG)))
        n = stream_N(j)
        n[0].need = ("M%d" % j,)
        streams.append(((j, 3), n))
        jj = stream_J(sched, j, m1, give_b)
        jj[0].need = ("N%d" % j,)
        streams.append(((j, 4), jj))
        prev_mf = j
    sched.run(streams)
    sched.emit(I("s_nop 1", "nop"))
    if GROUP_PRIO:
        sched.emit(I("s_setprio 0", "salu"))
    return sched, bf


@with_group_layout
def selftest_group(pattern, seed=1, crafted=False):
    rng = random.Random(seed * 77 + len(pattern))
    sched, bf = build_group(pattern)
    R = 64
    wR = pow(7, (P - 1) // R, P)
    tw = [pow(wR, i, P) for i in range(R // 2)]
    mats_addr = 0x10000
    mem = bytearray(mats_addr) + b"".join(twmat_bytes(t) for t in tw)
    st = Sim(mem, mats_addr, 0)
    choice = {}
    for (m0, m1, o1, o2, base) in bf:
        if o1 is not None:
            choice[o1], choice[o2] = rng.randrange(R // 2), rng.randrange(R // 2)
            st.operands[o1], st.operands[o2] = 2048 * choice[o1], 2048 * choice[o2]
    vals = [[rng.randrange(2**256) for _ in range(64)] for _ in range(4)]
    if crafted:
        for m in range(4):
            for l in range(64):
                c = rng.randrange(6)
                if c == 0:
                    vals[m][l] = 2**256 - 1 - rng.randrange(4)
                elif c == 1:
                    vals[m][l] = rng.randrange(4)
                elif c == 2:
                    vals[m][l] = P - 1 - rng.randrange(3)
    for m in range(4):
        for i in range(8):
            st.V[X(m, i)] = [(vals[m][l] >> (32 * i)) & M32 for l in range(64)]
    st.run(sched)
    ref = [list(v) for v in vals]
    for (m0, m1, o1, o2, base) in bf:
        for l in range(64):
            w = 1 if o1 is None else tw[choice[o2 if l >= 32 else o1]]
            a, b = ref[m0][l], ref[m1][l]
            ref[m0][l] = (a + b) % P
            ref[m1][l] = (a - b) * w % P
    bad = sum(1 for m in range(4) for l in range(64)
              if sum(st.V[X(m, i)][l] << (32 * i) for i in range(8)) % P != ref[m][l])
    return bad, sched


@with_group_layout
def emit_groups(path):
    """mfma_group.inc: the group blocks by pattern"""
    clob = ['"v%d"' % i for i in range(NTEMP)] + ['"s%d"' % i for i in range(S_CA, S_LAST)] + ['"vcc"', '"scc"']
    L = ["// GENERATED by gen_bflyasm.py -- do not edit.  Register groups of the LDS-resident tile pass with matrix-core butterflies:",
         "// x[h] pinned to v[96 + 8 h : 103 + 8 h]; v0..v%d, s%d..s%d, vcc and scc clobbered; o<k>lo / o<k>hi = byte offsets of the TwMat" % (NTEMP - 1, S_CA, S_LAST - 1),
         "// images (mfma_tw.cuh) of the k-th MFMA butterfly for lanes 0..31 / 32..63.",
         "#define SHK_GROUP_CLOBBERS " + ", ".join(clob)]
    stats = []
    for name in ("MMMM", "AMAA", "AA", "MM"):
        sched, bf = build_group(name)
        n = sum(i.nslots for i in sched.out if i.kind not in ("nop", "label"))
        nm = sum(1 for b in bf if b[2] is not None)
        stats.append((name, n, sched.nops))
        args = "shk_x8 (&x)[4], const shk_v16i& offs, uint32_t lane16, uint32_t mlo, uint32_t mhi" + \
            "".join(", uint32_t o%dlo, uint32_t o%dhi" % (k, k) for k in range(nm))
        L.append("// pattern %s: %d instruction slots, %d s_nop" % (name, n, sched.nops))
        L.append("__device__ __forceinline__ void shk_group_asm_%s(%s) {" % (name, args))
        L.append("  asm volatile(")
        for t in asm_text(sched):
            L.append('      "%s\\n\\t"' % t)
        L.append("      : " + ", ".join('"+{v[%d:%d]}"(x[%d])' % (DATA0 + 8 * m, DATA0 + 8 * m + 7, m) for m in range(4)))
        ins = '[offs] "v"(offs), [lane16] "v"(lane16), [mlo] "s"(mlo), [mhi] "s"(mhi)' + \
            "".join(', [o%dlo] "s"(o%dlo), [o%dhi] "s"(o%dhi)' % (k, k, k, k) for k in range(nm))
        L.append("      : " + ins)
        L.append("      : SHK_GROUP_CLOBBERS);")
        L.append("}")
    with open(path, "w") as fh:
        fh.write("\n".join(L) + "\n")
    return stats


def emit_group4(path):
    """EXPERIMENT (tools/ilp/group_bench.hip, not part of the library): one register group of an LDS-resident tile -- four
    elements per thread pinned to v[96:127], two levels = four MFMA butterflies, one fragment buffer, 70 temporaries -- so that
    the kernel around it fits 128 VGPRs = four waves per SIMD: what would the butterflies cost at the VALU passes' occupancy?"""
    global DATA0, FRAG_B, NTEMP
    saved = (DATA0, FRAG_B, NTEMP)
    DATA0, FRAG_B, NTEMP = 96, FRAG, 70
    try:
        bf = [(0, 2, 0, 2048, S_BASE), (1, 3, 4096, 6144, S_BASE), (0, 1, 8192, 10240, S_BASE), (2, 3, 8192, 10240, S_BASE)]
        sched, _ = build_stage(1, 6, bf)
        n = sum(i.nslots for i in sched.out if i.kind not in ("nop", "label"))
        clob = ['"v%d"' % i for i in range(NTEMP)] + ['"s%d"' % i for i in range(S_CA, S_LAST)] + ['"vcc"', '"scc"']
        L = ["// GENERATED by gen_bflyasm.py --group4 (experiment).  %d instruction slots, %d s_nop for 4 butterflies" % (n, sched.nops),
             "typedef uint32_t shk_x8 __attribute__((ext_vector_type(8)));",
             "__device__ __forceinline__ void shk_group4_asm(shk_x8 (&x)[4], const shk_v16i& offs, uint32_t lane16, uint32_t mlo, uint32_t mhi, uint32_t rho) {",
             "  asm volatile("]
        for t in asm_text(sched):
            L.append('      "%s\\n\\t"' % t)
        L.append("      : " + ", ".join('"+{v[%d:%d]}"(x[%d])' % (DATA0 + 8 * m, DATA0 + 8 * m + 7, m) for m in range(4)))
        L.append('      : [offs] "v"(offs), [lane16] "v"(lane16), [mlo] "s"(mlo), [mhi] "s"(mhi), [rho] "s"(rho)')
        L.append("      : " + ", ".join(clob) + ");")
        L.append("}")
        with open(path, "w") as fh:
            fh.write("\n".join(L) + "\n")
        return n, sched.nops
    finally:
        DATA0, FRAG_B, NTEMP = saved


if __name__ == "__main__":
    if "--group4" in sys.argv:
        print("group of 4 butterflies: %d slots, %d s_nop" % emit_group4(sys.argv[sys.argv.index("--group4") + 1]))
        sys.exit(0)
    if "--selftest" in sys.argv:
        tot = 0
        for stage in (1, 2):
            for log_r in (5, 6, 7, 8):
                for crafted in (False, True):
                    b, _ = selftest(stage, log_r, crafted=crafted, verbose=True)
                    tot += b
        for name in GROUP_PATTERNS:
            for crafted in (False, True):
                b, sc = selftest_group(name, crafted=crafted)
                print("group %s%s: %d s_nop, mismatches %d" % (name, " (crafted)" if crafted else "", sc.nops, b))
                tot += b
        sys.exit(1 if tot else 0)
    here = os.path.dirname(os.path.abspath(__file__))
    for st in emit_inc(os.path.join(here, "mfma_bfly.inc")):
        print("stage %d R=2^%d: %d slots, %d s_nop, %d butterflies" % st)
    for st in emit_groups(os.path.join(here, "mfma_group.inc")):
        print("group %s: %d slots, %d s_nop" % st)
