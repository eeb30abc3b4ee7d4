"""Host-side checks of the matrix-core NTT experiments kept under tools/not_kept/mfma (not part of the product or of the test
suite since round 4): the operand images, the instruction-level simulator runs of the generated stages, the hybrid lane mapping.
    python3 -m pytest tools/not_kept/mfma/test_mfma_host.py      (regenerate the .inc files first: python3 gen_bflyasm.py)"""
import os
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
P = 2**256 - 2**32 * 351 + 1


def test_twiddle_matrix_images(tmp_path):
    """The MFMA operand images of the matrix-core NTT passes (csrc/mfma_tw.cuh:shk_build_twmat, host code): for every
    byte k of the multiplicand, the 32 signed digits stored for it sum to w * 256^k (and to -w * 256^k) modulo p, every
    digit fits an i8, and the (lane, byte) placement is the one the kernels assume (output row i = byte position rho(i))."""
    import subprocess
    src = os.path.join(HERE, "twmat_dump.cpp")
    exe = tmp_path / "twmat_dump"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "--offload-arch=gfx950", "-std=c++17", "-I",
                           os.path.join(ROOT, "starks_amd", "csrc"), "-I", HERE, src, "-o", str(exe)], stderr=subprocess.DEVNULL)
    ws = [1, P - 1, 2, 0x80, int("7f" * 32, 16), int("80" * 32, 16) % P, pow(7, (P - 1) // 256, P),
          pow(7, (P - 1) // (1 << 24), P), 0x0123456789abcdef << 190, P - 12345]
    out = subprocess.check_output([str(exe)] + ["%064x" % w for w in ws]).decode().split()
    assert len(out) == len(ws)

    def rho(i):
        return 16 * ((i >> 2) & 1) + (i & 3) + 4 * (i >> 3)

    for w, line in zip(ws, out):
        img = bytes.fromhex(line)
        assert len(img) == 2048
        for half, target in ((img[:1024], w), (img[1024:], (P - w) % P)):
            # lane = i + 32 h holds, at byte j, the digit of position rho(i) of the column kappa = 16 h + j
            digits = [[0] * 32 for _ in range(32)]  # [kappa][position]
            for lane in range(64):
                i, h = lane & 31, lane >> 5
                for j in range(16):
                    b = half[16 * lane + j]
                    digits[16 * h + j][rho(i)] = b - 256 if b >= 128 else b
            assert sorted(rho(i) for i in range(32)) == list(range(32))
            for kappa in range(32):
                val = sum(d << (8 * m) for m, d in enumerate(digits[kappa]))
                assert (val - target * 256 ** kappa) % P == 0, (hex(w), kappa)
                assert -128 * ((256**32 - 1) // 255) <= val <= 127 * ((256**32 - 1) // 255)



def test_generated_mfma_stages_simulate_correctly():
    """The scheduled asm stages of the matrix-core tile pass (csrc/gen_bflyasm.py -> mfma_bfly.inc) run in the generator's own
    instruction-level simulator (64 lanes, MFMA operand layout, carries, the out-of-line rare-carry blocks) and must equal
    big-integer butterflies (a, b) -> (a + b, (a - b) w) on random and on crafted inputs that take the rare blocks; the
    committed .inc must be what the generator emits now."""
    import importlib.util
    path = os.path.join(HERE, "gen_bflyasm.py")
    spec = importlib.util.spec_from_file_location("gen_bflyasm", path)
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    taken = 0
    for stage, log_r, crafted in ((1, 7, False), (1, 5, True), (2, 8, True), (2, 6, False)):
        bad, sched = g.selftest(stage, log_r, crafted=crafted)
        assert bad == 0, (stage, log_r, crafted)
        assert sched.nops * 7 < sum(i.nslots for i in sched.out), "the schedule lost its interleaving"
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "mfma_bfly.inc")
        g.emit_inc(out)
        assert open(out).read() == open(os.path.join(HERE, "mfma_bfly.inc")).read(), \
            "mfma_bfly.inc is stale: run python3 starks_amd/csrc/gen_bflyasm.py"
    # the register groups of the LDS-resident tile (mfma_group.inc): every pattern, twiddles chosen per half-wave
    for name in g.GROUP_PATTERNS:
        for crafted in (False, True):
            bad, sched = g.selftest_group(name, crafted=crafted)
            assert bad == 0, (name, crafted)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "mfma_group.inc")
        g.emit_groups(out)
        assert open(out).read() == open(os.path.join(HERE, "mfma_group.inc")).read(), \
            "mfma_group.inc is stale: run python3 starks_amd/csrc/gen_bflyasm.py"



def test_hybrid_lane_mapping_is_a_bijection_with_shared_twiddles():
    """The thread -> row mapping of the matrix-core groups of the hybrid tile pass (csrc/ntt_mfma.hip:HybridLane::ibase), restated:
    for every tile shape and every group the rule selects, the 4 elements of all threads cover the R x T tile exactly once, and
    the 32 lanes of a half-wave agree on the low beta bits of their rows -- the bits a level-q twiddle (q <= beta + 1) depends on."""
    for tile_log in (10, 11):
        for log_r in range(5, 12):
            log_t = tile_log - log_r
            if log_t < 0 or log_t > 5:
                continue
            log_w = log_r + log_t - 8
            for g in range((log_r + 1) // 2):
                beta = log_r - 2 * (g + 1)
                if not (0 < beta <= 1 + log_w):
                    continue
                seen = set()
                for tid in range(1 << (log_r + log_t - 2)):
                    lane, wave = tid & 63, tid >> 6
                    lane_i = (lane & 31) >> log_t
                    u = (lane >> 5) | (wave << 1)
                    hi = lane_i | ((u >> beta) << (5 - log_t))
                    ibase = (u & ((1 << beta) - 1)) | (hi << (beta + 2))
                    # the half-wave's shared bits come from (lane bit 5, wave) only
                    assert ibase & ((1 << beta) - 1) == u & ((1 << beta) - 1)
                    for h in range(4):
                        i = ibase | (h << beta)
                        assert i < (1 << log_r)
                        seen.add((i, tid & ((1 << log_t) - 1)))
                assert len(seen) == 1 << (log_r + log_t), (tile_log, log_r, g)

