#!/usr/bin/env python3
"""Generates tools/ilp/ilp_bench.hip: issue-latency microbenchmarks for the instruction patterns of the MFMA butterfly
stages (dependent v_mad_u64_u32 chains at ILP 1..4, carry chains through SGPR pairs with fillers, swap chains), timed with
s_memtime in wave 0 of every block at 1 and 2 waves per SIMD."""
import os
REP = 64

def body_mad(ilp):
    L = []
    for r in range(REP):
        for c in range(ilp):
            L.append("v_mad_u64_u32 v[%d:%d], vcc, v%d, s20, v[%d:%d]" % (2 * c, 2 * c + 1, 20 + c, 2 * c, 2 * c + 1))
    return L, REP * ilp

def body_addc(fill):
    # one carry chain through s[22:23] with `fill` independent xors between links
    L = []
    for r in range(REP):
        L.append("v_addc_co_u32_e64 v0, s[22:23], v0, v20, s[22:23]")
        for f in range(fill):
            L.append("v_xor_b32_e32 v%d, s20, v%d" % (4 + f, 4 + f))
    return L, REP * (1 + fill)

def body_addc2(fill):
    # two carry chains (s[22:23], s[24:25]) alternating, `fill` xors after each pair
    L = []
    for r in range(REP):
        L.append("v_addc_co_u32_e64 v0, s[22:23], v0, v20, s[22:23]")
        L.append("v_addc_co_u32_e64 v1, s[24:25], v1, v21, s[24:25]")
        for f in range(fill):
            L.append("v_xor_b32_e32 v%d, s20, v%d" % (4 + f, 4 + f))
    return L, REP * (2 + fill)

def body_addc3():
    L = []
    for r in range(REP):
        L.append("v_addc_co_u32_e64 v0, s[22:23], v0, v20, s[22:23]")
        L.append("v_addc_co_u32_e64 v1, s[24:25], v1, v21, s[24:25]")
        L.append("v_addc_co_u32_e64 v2, s[26:27], v2, v22, s[26:27]")
    return L, REP * 3

def body_xor_indep():
    L = []
    for r in range(REP):
        for f in range(4):
            L.append("v_xor_b32_e32 v%d, s20, v%d" % (4 + f, 4 + f))
    return L, REP * 4

def body_xor_dep():
    return ["v_xor_b32_e32 v4, s20, v4"] * (REP * 4), REP * 4

def body_mad_indep():
    L = []
    for r in range(REP):
        for c in range(4):
            L.append("v_mad_u64_u32 v[%d:%d], vcc, v%d, s20, v[%d:%d]" % (2 * c, 2 * c + 1, 20 + c, 8 + 2 * c, 9 + 2 * c))
    return L, REP * 4

def body_swap():
    L = []
    for r in range(REP):
        L.append("v_permlane32_swap_b32_e32 v0, v1")
        L.append("v_permlane32_swap_b32_e32 v2, v3")
        L.append("v_permlane32_swap_b32_e32 v4, v5")
        L.append("v_permlane32_swap_b32_e32 v6, v7")
    return L, REP * 4

def body_norm():
    # the normalisation of one butterfly (two chains), as gen_bflyasm emits it
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "starks_amd", "csrc"))
    import gen_bflyasm as g
    L = []
    for r in range(8):
        for ins in g.stream_N(0):
            L += [t.replace("s75", "s20").replace("s76", "s20").replace("s77", "s20") for t in ins.text]
    return L, 8 * 32

def body_mad_addmix():
    # 2 mad chains + independent full-rate fillers 1:1
    L = []
    for r in range(REP):
        L.append("v_mad_u64_u32 v[0:1], vcc, v20, s20, v[0:1]")
        L.append("v_xor_b32_e32 v4, s20, v4")
        L.append("v_mad_u64_u32 v[2:3], vcc, v21, s20, v[2:3]")
        L.append("v_xor_b32_e32 v5, s20, v5")
    return L, REP * 4

def body_mfma_b2b():
    # 4 MFMAs back to back, then 60 independent VALU (mad ILP4)
    L = []
    for r in range(4):
        L += ["v_mfma_i32_32x32x32_i8 v[32:47], v[24:27], v[28:31], v[32:47]", "v_mfma_i32_32x32x32_i8 v[48:63], v[24:27], v[28:31], v[48:63]",
              "v_mfma_i32_32x32x32_i8 v[32:47], v[24:27], v[28:31], v[32:47]", "v_mfma_i32_32x32x32_i8 v[48:63], v[24:27], v[28:31], v[48:63]"]
        for k in range(15):
            for c in range(4):
                L.append("v_mad_u64_u32 v[%d:%d], vcc, v%d, s20, v[%d:%d]" % (2 * c, 2 * c + 1, 20 + c, 2 * c, 2 * c + 1))
    return L, 4 * 64

def body_mfma_spread():
    L = []
    for r in range(4):
        for q in range(4):
            L.append("v_mfma_i32_32x32x32_i8 v[%d:%d], v[24:27], v[28:31], v[%d:%d]" % (32 + 16 * (q & 1), 47 + 16 * (q & 1), 32 + 16 * (q & 1), 47 + 16 * (q & 1)))
            for k in range(15):
                c = k % 4
                L.append("v_mad_u64_u32 v[%d:%d], vcc, v%d, s20, v[%d:%d]" % (2 * c, 2 * c + 1, 20 + c, 2 * c, 2 * c + 1))
    return L, 4 * 64

def body_branch(every):
    # a not-taken s_cmp / s_cbranch pair every `every` VALU instructions
    L = []
    n = 0
    for r in range(REP * 4 // every):
        for k in range(every):
            L.append("v_xor_b32_e32 v%d, s20, v%d" % (4 + k % 4, 4 + k % 4))
            n += 1
        L.append("s_cmp_lg_u64 s[22:23], 0")
        L.append("s_cbranch_scc1 SKIP%d_%%=" % r)
        L.append("SKIP%d_%%=:" % r)
        n += 2
    return L, n

def body_salu(every):
    L = []
    n = 0
    for r in range(REP * 4 // every):
        for k in range(every):
            L.append("v_xor_b32_e32 v%d, s20, v%d" % (4 + k % 4, 4 + k % 4))
            n += 1
        L.append("s_add_u32 s28, s28, 1")
        L.append("s_addc_u32 s29, s29, 0")
        n += 2
    return L, n

TESTS = [("mad ILP1", body_mad(1)), ("mad ILP2", body_mad(2)), ("mad ILP3", body_mad(3)), ("mad ILP4", body_mad(4)),
         ("mad independent x4", body_mad_indep()), ("mad ILP2 + xor 1:1", body_mad_addmix()),
         ("addc chain +2 xor", body_addc(2)), ("addc chain +3 xor", body_addc(3)), ("2 addc chains +1 xor", body_addc2(1)),
         ("2 addc chains +2 xor", body_addc2(2)), ("3 addc chains", body_addc3()),
         ("xor independent x4", body_xor_indep()), ("xor dependent", body_xor_dep()), ("swap x4 indep", body_swap()),
         ("norm (2 chains, as generated)", body_norm()),
         ("4 MFMA back-to-back + 60 mad", body_mfma_b2b()), ("4 x (MFMA + 15 mad)", body_mfma_spread()),
         ("xor x20 + not-taken branch", body_branch(20)), ("xor x8 + not-taken branch", body_branch(8)),
         ("xor x8 + s_add/s_addc", body_salu(8))]

def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = ['// GENERATED by gen_ilp.py', '#include <hip/hip_runtime.h>', '#include <stdio.h>', '#include <stdint.h>', '#include <vector>', '#include <algorithm>']
    for k, (name, (lines, n)) in enumerate(TESTS):
        out.append("__global__ void __launch_bounds__(256) t%d(unsigned long long* o, int iters) {" % k)
        out.append("  unsigned long long t0 = 0, t1 = 0;")
        out.append('  asm volatile("s_mov_b32 s20, 0x101\\n\\ts_mov_b64 s[22:23], 0\\n\\ts_mov_b64 s[24:25], 0\\n\\ts_mov_b64 s[26:27], 0\\n\\t"')
        out.append('    "v_mov_b32 v67, 0\\n\\tv_mov_b32 v69, 0\\n\\t" ::: "s20","s22","s23","s24","s25","s26","s27","v67","v69");')
        out.append("  t0 = __builtin_amdgcn_s_memtime();")
        out.append("  for (int it = 0; it < iters; ++it) {")
        out.append("    asm volatile(")
        for ln in lines:
            out.append('      "%s\\n\\t"' % ln)
        clob = ", ".join('"v%d"' % i for i in range(80)) + ', "vcc", "s20","s22","s23","s24","s25","s26","s27","s28","s29","scc"'
        out.append("      ::: %s);" % clob)
        out.append("  }")
        out.append("  t1 = __builtin_amdgcn_s_memtime();")
        out.append("  if ((threadIdx.x & 63) == 0) o[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;")
        out.append("}")
    out.append("int main() {")
    out.append("  unsigned long long* d; hipMalloc(&d, 8 * 4 * 4096);")
    out.append("  std::vector<unsigned long long> h(4 * 4096);")
    out.append("  const int iters = 50;")
    for k, (name, (lines, n)) in enumerate(TESTS):
        out.append("  for (int w = 1; w <= 4; w *= 2) {")
        out.append("    int blocks = 256 * w;")
        out.append("    hipLaunchKernelGGL(t%d, dim3(blocks), dim3(256), 0, 0, d, iters); hipLaunchKernelGGL(t%d, dim3(blocks), dim3(256), 0, 0, d, iters);" % (k, k))
        out.append("    hipDeviceSynchronize(); hipMemcpy(h.data(), d, 8 * 4 * blocks, hipMemcpyDeviceToHost);")
        out.append("    std::sort(h.begin(), h.begin() + 4 * blocks);")
        out.append('    printf("%%-34s waves/SIMD %%d: %%6.2f cycles per instruction (median wave; %d instr per iteration)\\n", "%s", w, (double)h[2 * blocks] / (iters * %d.0));' % (n, name, n))
        out.append("  }")
    out.append("  return 0;")
    out.append("}")
    open(os.path.join(here, "ilp_bench.hip"), "w").write("\n".join(out) + "\n")

if __name__ == "__main__":
    main()
