"""CPU-only checks of the host logic: the C-ABI library loads and exports every symbol the header
declares (no compute without a GPU), the boundary element type, proof (un)packing and compression."""
import os
import re

import pytest

from conftest import ROOT, load_golden

P = 2**256 - 2**32 * 351 + 1


def _build():
    import __graft_entry__ as ge
    ge.build()


def test_library_exports_every_declared_symbol():
    _build()
    from starks_amd import _lib
    header = open(os.path.join(ROOT, "include", "starkhip.h")).read()
    declared = sorted(set(re.findall(r"\b(sh_[a-z_0-9]+)\s*\(", header)))
    assert declared == _lib.exported_symbols()  # lib() getattr()s each one: a missing export raises
    assert _lib.lib().sh_version().startswith(b"starkhip")
    assert _lib.lib().sh_strerror(-2) == b"root_of_unity does not have order n"
    assert _lib.lib().sh_fri_proof_len(512, 512, 40) == load_golden("fri.json")[0]["flat_len"]


def test_no_cpu_fallback_without_gpu():
    _build()
    from starks_amd import _lib, fft, IntegersModP
    if _lib.lib().sh_device_count() > 0:
        pytest.skip("a GPU is present")
    F = IntegersModP(P)
    with pytest.raises(_lib.StarkHipError):
        fft.fft_1d(F, [1, 2, 3], P, F(7) ** ((P - 1) // 8))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "starks_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "liboracle" not in src and "pyoracle" not in src.replace("see pyoracle", ""), f


def test_field_element_semantics():
    """modp.py call surface incl. the reference's KATs (test_modpy.py:28-35,54-61)."""
    from starks_amd import IntegersModP
    F = IntegersModP(P)
    assert F is IntegersModP(P)
    assert F(2**256) == F(2**32 * 351 - 1) and F(5) != F(11)
    assert int(F(7) ** ((P - 1) // 512)) == pow(7, (P - 1) // 512, P)
    assert F(b"\xff" * 32).n == 2**256 - 1  # bytes ctor does not reduce
    assert (F(b"\xff" * 32) * 1).n == (2**256 - 1) % P
    assert F(3) * 5 == 15 and 5 * F(3) == 15 and 1 - F(2) == P - 1 and -F(1) == P - 1
    assert F(3) / F(3) == 1 and 1 / F(2) * 2 == 1 and F(5).inverse() * 5 == 1
    assert F(5).to_bytes() == (5).to_bytes(32, "big")
    m7 = IntegersModP(7)
    assert m7(5) == 1 / m7(3) and m7(0) == m7(3) + m7(4)
    g = load_golden("field.json")
    for c in g["cases"][:50]:
        a, b = F(int(c["a"], 16)), F(int(c["b"], 16))
        assert (a + b).to_bytes().hex() == c["add"] and (a - b).to_bytes().hex() == c["sub"] and (a * b).to_bytes().hex() == c["mul"]


def test_unpack_and_compress_against_golden():
    from starks_amd import fri, compression
    rec = [r for r in load_golden("fri.json") if r["name"] == "fri_deg512"][0]
    flat = open(os.path.join(ROOT, "tests", "golden", "fri_deg512.flat.bin"), "rb").read()
    proof = fri.unpack_proof(flat, 512, 512, 40)
    assert len(proof) == 4 and [len(b) for b in proof[0][1][0]] == rec["branch_lens"][0]
    c = compression.compress_fri(proof)
    assert compression.decompress_fri(c) == proof
    assert compression.proof_bytes(c) == open(os.path.join(ROOT, "tests", "golden", "fri_deg512.proof.bin"), "rb").read()
    assert compression.bin_length(c) == rec["proof_bytes_len"]
    g = load_golden("compression.json")
    br = [[bytes.fromhex(x) for x in b] for b in g["branches"]]
    assert [x.hex() for x in compression.compress_branches(br)] == g["compressed"]
    # host verifier accepts the reference-generated proof and rejects a corrupted one
    assert fri.verify_low_degree_proof(proof, bytes.fromhex(rec["eval_root"]), int(rec["w"], 16), 512)
    bad = [[proof[0][0], [[list(b) for b in bs] for bs in proof[0][1]]]] + proof[1:]
    bad[0][1][3][0][0] = b"\x01" * 32
    with pytest.raises(AssertionError):
        fri.verify_low_degree_proof(bad, bytes.fromhex(rec["eval_root"]), int(rec["w"], 16), 512)


def test_host_index_sampling_matches_golden():
    from starks_amd.utils import get_pseudorandom_indices
    for c in load_golden("utils.json")["pseudorandom_indices"]:
        assert get_pseudorandom_indices(bytes.fromhex(c["entropy"]), c["modulus"], c["count"], c["exclude"]) == c["out"]


def test_stark_helpers_restated_from_text():
    """stark.py:106-126, 390-402 (un-importable module): structure checks on the restatement."""
    import hashlib
    from starks_amd.stark import get_pseudorandom_ks, compute_merkle_spot_checks
    from starks_amd.merkle_tree import verify_branch
    root = hashlib.blake2s(b"r").digest()
    ks = get_pseudorandom_ks(root, 4)
    assert ks == [int.from_bytes(hashlib.blake2s(root + s).digest(), "big") for s in (b"0x01", b"0x02", b"0x03", b"0x04")]
    assert get_pseudorandom_ks(root, 6)[0] == int.from_bytes(hashlib.blake2s(root + b"0x00").digest(), "big")
    assert get_pseudorandom_ks(root, 12) is None
    # spot checks on host-built trees (heap layout of merkle_tree.py:36-56)
    def tree(vals):
        n = len(vals)
        q = n // 4
        nodes = [b""] * n + [vals[i + j * q] for i in range(q) for j in range(4)]
        for i in range(n - 1, 0, -1):
            nodes[i] = hashlib.blake2s(nodes[2 * i] + nodes[2 * i + 1]).digest()
        return nodes
    n, ext = 64, 8
    m = tree([(i * 7 + 1).to_bytes(32, "big") for i in range(n)])
    lt = tree([(i * 11 + 3).to_bytes(32, "big") for i in range(n)])
    br = compute_merkle_spot_checks(m, lt, n, ext, samples=10)
    assert len(br) == 30
    from starks_amd.utils import get_pseudorandom_indices
    pos = get_pseudorandom_indices(lt[1], n, 10, exclude_multiples_of=ext)
    assert all(p % ext for p in pos)
    for k, p in enumerate(pos):
        assert verify_branch(m[1], p, br[3 * k], output_as_int=True) == p * 7 + 1
        assert verify_branch(m[1], (p + ext) % n, br[3 * k + 1], output_as_int=True) == ((p + ext) % n) * 7 + 1
        assert verify_branch(lt[1], p, br[3 * k + 2], output_as_int=True) == p * 11 + 3


def test_stark_host_mirror_on_reference_proofs():
    """The host side of STARK (step-polynomial type, witness helpers, proof unpacking, verifier) against the proofs of
    the live reference (tests/golden/stark.json via the oracle, which test_oracle_golden pins byte for byte)."""
    _build()
    import hashlib
    from oracle import pyoracle as po
    from starks_amd import IntegersModP, stark
    from starks_amd.air import AIR
    from starks_amd.multivariate_polynomial import multivariates_over, generate_Xi_s
    F = IntegersModP(P)
    X1, X2 = generate_Xi_s(F, 2)
    assert (X1 + X2**3).coefficients == {(1, 0): 1, (0, 3): 1} and (X1 + X2**3).degree() == 3
    assert (3 * X2 + F(2) * X1 - 1).coefficients == {(0, 1): 3, (1, 0): 2, (0, 0): P - 1}
    assert (X1 * X2 - X2 * X1).is_zero() and int((X1 + X2**3)([F(2), 5])) == 127
    for c in load_golden("stark.json"):
        mv = multivariates_over(F, c["width"]).factory
        polys = [mv({tuple(k): v for k, v in d}) for d in c["step_polys"]]
        sp = [{tuple(k): v for k, v in d} for d in c["step_polys"]]
        air = AIR(F, c["width"], c["inputs"], c["steps"], polys, c["ext"])
        witness = air.generate_witness()
        assert [[int(v) for v in col] for col in witness] == [[int(x, 16) for x in col] for col in c["witness"]]
        boundary = air.generate_boundary_constraints()
        assert [(b[0], b[1], int(b[2])) for b in boundary] == [(0, j, v % P) for j, v in enumerate(c["inputs"])]
        S = stark.STARK(F, c["steps"], c["ext"], c["width"], polys)
        assert S.get_degree() == c["degree"] == stark.pack_step_polys(polys, c["width"])[3]
        ref = po.mk_stark_proof([[int(v) for v in col] for col in witness], c["inputs"], sp, c["steps"], c["ext"])
        flat = po.stark_flat(ref)
        assert hashlib.sha256(flat).hexdigest() == c["flat_sha"]
        assert stark.proof_len(c["steps"], c["ext"], c["width"], c["degree"]) == len(flat)   # host-only C entry point
        assert stark.unpack_proof(flat, c["steps"], c["ext"], c["width"], c["degree"]) == ref
        assert S.verify_proof(ref, witness, boundary)
        bad = [ref[0], ref[1], [list(b) for b in ref[2]], ref[3]]
        leaf = bytearray(bad[2][0][0])
        leaf[40] ^= 1
        bad[2][0][0] = bytes(leaf)
        with pytest.raises(AssertionError):
            S.verify_proof(bad, witness, boundary)


def test_reference_fft_unit_tests_on_other_fields():
    """The reference's own FFT unit tests on Z/31 with a 6th root of unity (test_fft.py:98-113, 132-149, 151-168): not
    the accelerated field, so `fft_1d` follows the reference's recursion on the host; values from the live reference
    (tests/golden/ntt.json:mod31_n6)."""
    from starks_amd import IntegersModP
    from starks_amd.fft import NonBinaryFFT, fft_1d
    from starks_amd.polynomial import polynomials_over
    g = load_golden("ntt.json")["mod31_n6"]
    F = IntegersModP(31)
    polys = polynomials_over(F).factory
    poly = polys([0, 1, 2, 3])
    w = F(3) ** ((31 - 1) // 6)
    solver = NonBinaryFFT(F, w)
    ev = solver.fft(poly)
    assert len(ev) == 6 and [int(v) for v in ev] == g["fwd"] and all(isinstance(v, F) for v in ev)
    assert solver.inv_fft(ev) == poly
    assert [int(v) for v in fft_1d(F, [0, 1, 2, 3], 31, w)] == g["fwd"]


def test_generated_asm_includes_are_current(tmp_path):
    """csrc/fp256_mulasm.inc, fp256_addasm.inc and blake2s_asm.inc are committed outputs of gen_mulasm.py / gen_addasm.py (whose
    check() asserts the carry wait states of every generated column) and gen_blake2s_asm.py (which executes its own operand
    wiring against a plain G-by-G BLAKE2s before writing): regenerating them must reproduce the committed files byte for byte."""
    import shutil, subprocess, sys
    csrc = os.path.join(ROOT, "starks_amd", "csrc")
    for gen, inc in (("gen_mulasm.py", "fp256_mulasm.inc"), ("gen_addasm.py", "fp256_addasm.inc"),
                     ("gen_blake2s_asm.py", "blake2s_asm.inc")):
        shutil.copy(os.path.join(csrc, gen), tmp_path / gen)
        subprocess.check_call([sys.executable, str(tmp_path / gen)])
        assert (tmp_path / inc).read_bytes() == open(os.path.join(csrc, inc), "rb").read(), inc


def test_pair_constant_product_on_the_host(tmp_path):
    """fp_mul2 (csrc/fp256.cuh: the product by a table constant kept as the pair (w, w 2^128 mod p), what every NTT
    butterfly uses) against fp_mul through the portable C paths of the same header: 400 k random and edge operands, canonical
    and lazily reduced second images.  The device's inline-asm path is pinned by the GPU parity tests and lab/r01_r03/mul2_bench.hip."""
    import subprocess
    src = os.path.join(ROOT, "tests", "native", "mul2_host.cpp")
    exe = tmp_path / "mul2_host"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-std=c++17", "-I",
                           os.path.join(ROOT, "starks_amd", "csrc"), src, "-o", str(exe)], stderr=subprocess.DEVNULL)
    out = subprocess.check_output([str(exe)]).decode()
    assert "400000 products, 0 mismatches" in out


def test_prover_input_shape_errors():
    """Short witness / input buffers and short boundary lists are refused on the host (the C side would read past them)."""
    from starks_amd import IntegersModP, stark
    from starks_amd.multivariate_polynomial import generate_Xi_s
    F = IntegersModP(P)
    X1, X2 = generate_Xi_s(F, 2)
    polys = [X1, X1 + X2**3]
    with pytest.raises(ValueError):
        stark.prove_flat(bytes(32 * 2 * 8 - 32), bytes(64), 8, 8, 2, polys)
    with pytest.raises(ValueError):
        stark.prove_flat(bytes(32 * 2 * 8), bytes(32), 8, 8, 2, polys)
    with pytest.raises(ValueError):
        stark.prove_flat(bytes(32 * 2 * 8), bytes(64), 8, 8, 2, polys, batch=2)
    S = stark.STARK(F, 8, 8, 2, polys)
    with pytest.raises(IndexError):   # the reference fails at boundary[dim] (stark.py:91)
        S.mk_proof([[1] * 8, [2] * 8], [(0, 0, 1)])
    with pytest.raises(ValueError):
        S.mk_proof([[1] * 8], [(0, 0, 1), (0, 1, 2)])


def test_c_abi_from_plain_c(tmp_path):
    """include/starkhip.h is C99 and a plain C program can link the library and call its host-only entry points."""
    import subprocess
    _build()
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "starkhip.h"
int main(void) {
  if (strncmp(sh_version(), "starkhip", 8)) return 1;
  if (strcmp(sh_strerror(SH_ERR_CONSTRAINT), "the witness violates a transition constraint")) return 2;
  if (sh_fri_proof_len(512, 512, 40) == 0) return 3;
  if (sh_stark_proof_len(8, 8, 2, 3, 80) != 147808) return 4;   /* tests/golden/stark.json: mimc_w2_s8 */
  if (sh_stark_proof_len(8, 8, 10, 3, 80) != 0) return 5;
  if (sh_ntt(NULL, NULL, 0, NULL, 8, NULL, 0) != SH_ERR_INVALID) return 6;
  if (sh_ntt_passes(1u << 20, 1) != 2 || sh_ntt_passes(1u << 20, 8) != 2 || sh_ntt_passes(1u << 19, 64) != 2 || sh_ntt_passes(1u << 24, 1) != 3 ||
      sh_ntt_passes(256, 1) != 1 || sh_ntt_passes(6, 1) != 0) return 7;
  if (sh_ctx_trim(NULL) != SH_ERR_INVALID || sh_stark_status_batch(NULL, NULL, 0) != SH_ERR_INVALID) return 8;
  printf("ok\n");
  return 0;
}
''')
    exe = tmp_path / "abi"
    libdir = os.path.join(ROOT, "starks_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", libdir, "-lstarkhip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.check_output([str(exe)]).strip() == b"ok"


def test_host_generality_outside_the_hot_path():
    """What the device code cannot represent stays available through the reference's call sites, on the host: merkelize of
    leaves that are not 32-byte values or not a power-of-two count (merkle_tree.py:36-56 hashes whatever to_bytes gives),
    mul_polys over another field (fft.py:334-345).  Checked against hashlib / schoolbook arithmetic -- no GPU, no oracle."""
    import random
    from hashlib import blake2s
    from starks_amd import merkle_tree as mt
    from starks_amd import fft
    from starks_amd.modp import IntegersModP
    rng = random.Random(9)
    for n in (1, 3, 4, 6, 7, 12, 16):
        L = [rng.randbytes(rng.choice([5, 32, 40])) for _ in range(n)]
        nodes = mt.merkelize(L)
        q = n // 4
        perm = [L[i + j * q] for i in range(q) for j in range(4)]
        m = len(perm)
        assert len(nodes) == 2 * m and nodes[m:] == perm
        for i in range(1, m):
            assert nodes[i] == blake2s(nodes[2 * i] + nodes[2 * i + 1]).digest()
    F = IntegersModP(31)
    for root, order in ((2, 5), (6, 6), (15, 10), (3, 30)):
        a = [rng.randrange(31) for _ in range(order // 2)]
        b = [rng.randrange(31) for _ in range(order - order // 2)]
        got = [int(x) for x in fft.mul_polys([F(x) for x in a], [F(x) for x in b], F(root))]
        prod = [0] * order
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                prod[(i + j) % order] = (prod[(i + j) % order] + x * y) % 31
        assert got == [(order * v) % 31 for v in prod]   # n * (a * b): the reference omits the 1/n (fft.py:345)
    # get_power_cycle in another field (test_utils.py:20-30 walks a 6th root of unity mod 31), and of an odd order in the MiMC field
    from starks_amd.utils import get_power_cycle
    assert get_power_cycle(F(3) ** 5, F) == [1, 26, 25, 30, 5, 6]
    P = 2**256 - 2**32 * 351 + 1
    Fp = IntegersModP(P)
    g5 = Fp(7) ** ((P - 1) // 5)  # 5 divides p - 1 = 2^32 (2^224 - 351)
    cyc = get_power_cycle(g5, Fp)
    assert len(cyc) == 5 and int(cyc[4]) * int(g5) % P == 1


def test_proof_stream_round_trips_on_random_shapes():
    """compress_fri / compress_branches and their single-pass decoders (starks_amd/compression.py, format of
    compression.py:1-101) on random nested proofs with many repeated nodes (back-references, repeated markers)."""
    import random
    from starks_amd import compression as cz
    rng = random.Random(11)
    pool = [rng.randbytes(32) for _ in range(12)]          # few distinct nodes: most objects become 2-byte references
    for trial in range(60):
        layers = []
        for _ in range(rng.randrange(0, 4)):
            yproofs = [[[rng.choice(pool) for _ in range(rng.randrange(0, 5))] for _ in range(rng.randrange(0, 4))]
                       for _ in range(rng.randrange(1, 4))]
            layers.append([rng.choice(pool), yproofs])
        prf = layers + [[rng.choice(pool) for _ in range(rng.randrange(0, 6))]]
        c = cz.compress_fri(prf)
        assert cz.decompress_fri(c) == prf
        assert all(len(x) in (2, 4, 32) for x in c)
        branches = [[rng.choice(pool) for _ in range(rng.randrange(0, 6))] for _ in range(rng.randrange(0, 6))]
        assert cz.decompress_branches(cz.compress_branches(branches)) == branches
        assert cz.bin_length(c) == sum(len(x) + (1 if len(x) == 32 else 0) for x in c)


def test_launch_knobs_are_parsed_once_and_race_free(tmp_path):
    """csrc/knobs.hpp (every STARKHIP_* launch knob and the plan choice built on them) under ThreadSanitizer: eight host threads
    take their first look at the knobs at the same moment -- the way the first transforms of two contexts on two threads meet --
    and ask for the plan of every size; no data race, one object, identical answers.  Also: the environment is read once (a
    later change is not seen), malformed values fall back to the defaults."""
    import subprocess
    exe = tmp_path / "knobs_tsan"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-I",
                           os.path.join(ROOT, "starks_amd", "csrc"), os.path.join(ROOT, "tests", "native", "knobs_tsan.cpp"),
                           "-o", str(exe)])
    env = {k: v for k, v in os.environ.items() if not k.startswith("STARKHIP_")}
    env["TSAN_OPTIONS"] = "halt_on_error=1 exitcode=66"
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip() == "ok tile_log=10 big=11 logs0=0 swz=1 tw2=24 cache=-1 radices=0 passes20=2 passes24=3"
    env.update(STARKHIP_NTT_RADICES="7,7,6", STARKHIP_TILE_LOGS="11,9", STARKHIP_XCD_SWZ="7", STARKHIP_TW2_MAX_LOG="20",
               STARKHIP_TILE_LOG_BIG="13", STARKHIP_PLAN_CACHE_MB="64")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip() == "ok tile_log=10 big=11 logs0=11 swz=0 tw2=20 cache=64 radices=3 passes20=3 passes24=3"


def test_library_plan_queries_from_two_threads():
    """The shipped library: sh_ntt_passes (a launch-free entry that goes through the same knobs and plan choice as a
    transform) called from two host threads at once agrees with the single-threaded answers."""
    import ctypes, threading
    from starks_amd import _lib
    L = _lib.lib()
    want = [int(L.sh_ntt_passes(1 << lg, 1)) for lg in range(0, 29)]
    assert want[8] == 1 and want[16] == 2 and want[20] == 2 and want[24] == 3 and want[28] == 4
    got = [None, None]

    def work(i):
        got[i] = [[int(L.sh_ntt_passes(1 << lg, 1)) for lg in range(0, 29)] for _ in range(200)]

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(row == want for g in got for row in g)


def test_to_wire_accepts_iterators_without_losing_elements():
    """_lib.to_wire on a generator whose fast path fails part way (a field element after plain ints): the general path must see
    every value, not the tail the fast path left."""
    from starks_amd import _lib
    from starks_amd.modp import IntegersModP
    F = IntegersModP(P)
    vals = [1, 2, F(3), -1, 2**256 + 5]
    want = b"".join((int(v) % P if not 0 <= int(v) < 2**256 else int(v)).to_bytes(32, "big") for v in vals)
    assert _lib.to_wire(iter(vals)) == want
    assert _lib.to_wire(v for v in vals) == want
    assert _lib.to_wire(vals) == want and _lib.to_wire(tuple(vals)) == want


def test_ntt_tile_mapping_keeps_barrier_free_exchanges_inside_a_wave(tmp_path):
    """csrc/ntt_kernels.cuh on the host (tests/native/tile_map_host.cpp, hipcc): for every tile shape the library instantiates, each
    register group's threads partition the tile, the LDS image is a linear bijection, and wherever two consecutive groups hand their
    elements over WITHOUT a workgroup barrier the LDS slots a wave reads are exactly the ones the same wave wrote.  (Round 4: a
    Merkle kernel broke the analogous rule -- two hand-overs with different per-wave slices -- and only showed under a second
    stream; the Merkle side is a static_assert in kernels.hip and a two-context GPU test.)"""
    import subprocess
    exe = tmp_path / "tile_map_host"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "starks_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "tile_map_host.cpp"), "-o", str(exe)], stderr=subprocess.DEVNULL)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip() == "37 tile shapes, 85 barrier-free exchanges, 0 failures"


def test_every_barrier_free_lds_handover_stays_inside_its_waves_slice(tmp_path):
    """csrc/wave_chunks.cuh on the host (tests/native/wave_slices_host.cpp, g++): the hashing kernels take the constants of their
    per-wave LDS hand-overs from the plans in that header; for every plan, hand-over and wave, the LDS indices touched lie inside the
    wave's one slice and the array, and what a wave parks is what it moves.  The plan of commit 284afb6's Merkle mid kernel (slices of
    512 chunks on the load, 256 on the store: round 4's race, invisible to the single-stream GPU suite) is fed to the same checker and
    must be rejected."""
    import subprocess
    exe = tmp_path / "wave_slices_host"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "starks_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "wave_slices_host.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("0 failures, r04 bug caught"), out.stdout
    # the kernels really use those plans: no literal hand-over constants left in the kernel sources
    import re
    for f in ("kernels.hip", "stark.hip"):
        src = open(os.path.join(ROOT, "starks_amd", "csrc", f)).read()
        for m in re.finditer(r"wave_(?:store|load)_chunks<([^(]*)>\(", src):
            assert "handover_at<" in m.group(1), (f, m.group(0))


def test_dpp_reads_keep_their_distance_from_the_producer(tmp_path):
    """gfx950: a VALU write of a VGPR followed by a DPP read of it needs two wait states.  The compiler takes care of that in its own
    code, but not inside inline asm -- and the quad-lane BLAKE2s half-rounds (csrc/blake2s.cuh: B2Q_HALF) are asm blocks whose DPP
    reads rely on being at least that far behind the last writer, including whatever the compiler puts right in front of a block.
    Checked on the compiled ISA of kernels.hip (every kernel that hashes in the serial form): for every *_dpp instruction, none of the
    instructions inside the two preceding wait states writes its DPP source."""
    import re, subprocess
    asm = tmp_path / "kernels.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-S", "--cuda-device-only",
                           "-I", os.path.join(ROOT, "starks_amd", "csrc"), os.path.join(ROOT, "starks_amd", "csrc", "kernels.hip"),
                           "-o", str(asm)], stderr=subprocess.DEVNULL)
    ins = []
    for ln in open(asm):
        ln = ln.split(";")[0].strip()
        if ln and re.match(r"^[vs]_|^ds_|^global_|^buffer_|^scratch_|^flat_", ln):
            ins.append(ln)

    def regs(tok):
        tok = tok.strip()
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            return {int(m.group(1))}
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()

    checked = 0
    for i, ln in enumerate(ins):
        op = ln.split()[0]
        if not op.endswith("_dpp"):
            continue
        ops = [t for t in ln[len(op):].split(" quad_perm")[0].split(" row_")[0].split(",")]
        src = regs(ops[1])  # VOP1 / VOP2 DPP: vdst, src0 (the DPP operand), [src1]
        assert src, ln
        checked += 1
        waits, j = 0, i - 1
        while waits < 2 and j >= 0:
            prev = ins[j]
            pop = prev.split()[0]
            if pop == "s_nop":
                waits += int(prev.split()[1], 0) + 1
            else:
                if pop.startswith("v_") and not pop.startswith("v_cmp"):
                    dst = regs(prev[len(pop):].split(",")[0])
                    assert not (dst & src), "DPP hazard: '%s' reads what '%s' wrote %d instruction(s) earlier" % (ln, prev, i - j)
                waits += 1
            j -= 1
    assert checked >= 80 * 4  # 80 DPP reads per compression, in the top kernels (three leaf modes x two widths) and the sampling kernels


def test_wire_backed_sequences_keep_list_semantics():
    """starks_amd/wireseq.py: what the drop-in functions return instead of lists of 2^20 element objects.  Indexing (negative,
    slices, strides, reversal), iteration, `in`, equality against lists of ints / elements / each other, concatenation, the
    zero-copy hand-over to the next stage (_lib.to_wire), Polynomial's trailing-zero strip on the bytes, NodeList's b"" slot 0 and
    the packed-leaf tail."""
    from starks_amd.wireseq import WireList, NodeList
    from starks_amd.modp import IntegersModP
    from starks_amd._lib import MIMC_P as P, to_wire
    from starks_amd.polynomial import polynomials_over
    F = IntegersModP(P)
    vals = [5, 0, P - 1, 7, 0, 0, 9, 0]
    raw = b"".join(v.to_bytes(32, "big") for v in vals)
    w = WireList(raw, F)
    assert len(w) == 8 and w[2] == P - 1 and isinstance(w[0], F) and int(w[-2]) == 9 and w[3] + 1 == 8
    with pytest.raises(IndexError):
        w[8]
    assert w == vals and w == [F(v) for v in vals] and w == tuple(vals) and not (w != vals) and w != vals[:-1] and w != [1] * 8
    assert w[1:3] == [0, P - 1] and w[::2] == [5, P - 1, 0, 9] and w[::-1] == vals[::-1] and w[5:2] == [] and isinstance(w[::2], WireList)
    assert list(w) == [F(v) for v in vals] and 7 in w and 8 not in w and (w + [1])[-1] == 1 and ([1] + w)[0] == 1
    assert to_wire(w) is raw and to_wire(w[2:4]) == raw[64:128]  # the next stage takes the bytes as they are
    assert w == WireList(bytes(raw), F) and w.ints() == vals and w.tolist() == vals
    big = WireList(b"".join(i.to_bytes(32, "big") for i in range(10000)), F)  # iteration crosses its 4096-value blocks
    assert [int(v) for v in big] == list(range(10000))
    poly = polynomials_over(F).factory(w)
    assert len(poly.coefficients) == 7 and poly.coefficients == vals[:7] and isinstance(poly.coefficients, WireList)
    assert poly(2) == sum(v * 2**i for i, v in enumerate(vals)) % P
    assert polynomials_over(F).factory(WireList(bytes(64), F)).is_zero()
    loose = WireList((P + 3).to_bytes(32, "big"), F, canonical=False)  # unreduced bytes: elementwise semantics, reduced on access
    assert loose == [3] and loose[0] == 3 and loose == WireList((3).to_bytes(32, "big"), F)
    nodes = NodeList(bytes(32) + b"\x01" * 32 + b"\x02" * 32 + b"\x03" * 32)
    assert len(nodes) == 4 and nodes[0] == b"" and nodes[1] == b"\x01" * 32 and nodes[-1] == b"\x03" * 32
    assert nodes[1:3] == [b"\x01" * 32, b"\x02" * 32] and nodes == [b"", b"\x01" * 32, b"\x02" * 32, b"\x03" * 32]
    assert nodes[len(nodes) // 2 + 1] == b"\x03" * 32  # mk_branch's arithmetic (merkle_tree.py:59-68) works on it unchanged
    t = NodeList(bytes(64), tail=b"\x07" * 96 * 2, tail_width=96)
    assert len(t) == 4 and t[2] == b"\x07" * 96 and t[0] == b"" and t[3] == b"\x07" * 96


def test_bench_launch_schedule_covers_every_unit_once():
    """bench.py's launch sizes (on the device: equal chunks; delivering: a halving tail, so that the one copy nothing hides is small):
    every unit exactly once, in order, for every shard size the N = 1 .. 8 runs produce."""
    import bench
    for units in (0, 1, 31, 32, 33, 64, 128, 171, 256, 512, 1000):
        for chunk in (32, 64, 128, 256):
            for deliver in (False, True):
                sch = bench.launch_schedule(units, chunk, deliver)
                assert [c for c, _ in sch] == [sum(k for _, k in sch[:i]) for i in range(len(sch))]
                assert sum(k for _, k in sch) == units and all(0 < k <= chunk for _, k in sch)
    assert [k for _, k in bench.launch_schedule(512, 256, True)] == [256, 128, 64, 32, 32]
    assert [k for _, k in bench.launch_schedule(512, 256, False)] == [256, 256]
    assert [k for _, k in bench.launch_schedule(64, 32, True)] == [32, 32]
