#!/usr/bin/env python3
"""Generates mathlib_amd/csrc/fp_mul_comba.inc: the gfx950 device body of fp_mul for N = 8 and N = 12
32-bit limbs.  Product-scanning (Comba) Montgomery multiplication: every column is accumulated in a
96-bit register triple by  v_mad_u64_u32 (+ its carry-out)  ->  v_addc_co_u32  pairs, so a 384-bit
product costs 288 + 288 vector instructions instead of the ~1 250 hipcc emits for the CIOS loop
(profiles/r01_ubench_int.txt has the issue costs).  One asm statement per column part keeps hipcc's
per-statement `s_nop` pad to ~50 per multiplication.  The modulus limbs are "s" (SGPR) operands.
The gfx950 "VALU writes SGPR/VCC -> VALU reads it" rule needs 2 wait states and nothing inside an asm string
is padded by hipcc; mac_block software-pipelines the v_addc one product behind its v_mad to satisfy it.

Run: python tools/gen_fp_comba.py > mathlib_amd/csrc/fp_mul_comba.inc
"""

MAXOPS = 30


def mac_block(pairs):
    """asm text for a list of (x_operand_index, y_operand_index) products into %0 (acc, 64-bit) / %1 (ext);
    %2 is a scratch SGPR pair.

    gfx950 hazard (LLVM GCNHazardRecognizer, VALUWriteSGPRVALURead = 2 wait states on the gfx940 family):
    a VALU that writes an SGPR / VCC must be followed by 2 wait states before a VALU reads it as carry-in,
    and hipcc pads nothing inside an asm string.  Schedule: carry-outs alternate between VCC (even products,
    consumed by the 4-byte VOP2 v_addc) and the SGPR pair %2 (odd products, VOP3 v_addc), and every v_addc
    is issued one product late:   mad_k  addc_{k-1}  mad_{k+1}  addc_k ...   so exactly two instructions
    separate a v_mad from the v_addc that reads its carry; the ends of a block are padded with s_nop.
    Measured (tools/ubench_carry.hip, profiles/r01_ubench_carry.txt and bench runs): unpadded adjacent pairs
    are ~9 % faster in k_accumulate but break the rule; `s_nop 1` after every mad is rule-conformant but
    15 % slower than unpadded at 2 waves/SIMD.
    """
    n = len(pairs)
    lines = []
    pos_of_mad = {}

    def addc(k):
        between = len(lines) - pos_of_mad[k] - 1
        if between < 2:
            lines.append("s_nop %d" % (2 - between - 1))
        if k % 2 == 0:
            lines.append("v_addc_co_u32 %1, vcc, 0, %1, vcc")
        else:
            lines.append("v_addc_co_u32 %1, %2, 0, %1, %2")

    for k in range(n + 1):
        if k < n:
            pos_of_mad[k] = len(lines)
            xi, yi = pairs[k]
            lines.append("v_mad_u64_u32 %%0, %s, %%%d, %%%d, %%0" % ("vcc" if k % 2 == 0 else "%2", xi, yi))
        if k - 1 >= 0:
            addc(k - 1)
    return "\\n\\t".join(lines)


def emit_stmt(out, prods):
    """prods: list of (x_expr, x_constraint, y_expr, y_constraint).  Splits to respect the operand limit."""
    per = (MAXOPS - 3) // 2
    for off in range(0, len(prods), per):
        chunk = prods[off : off + per]
        ops = []
        pairs = []
        for x, xc, y, yc in chunk:
            pairs.append((3 + len(ops), 4 + len(ops)))
            ops.append('"%s"(%s)' % (xc, x))
            ops.append('"%s"(%s)' % (yc, y))
        out.append('    asm("%s" : "+v"(acc), "+v"(ext), "=&s"(cy) : %s : "vcc");' % (mac_block(pairs), ", ".join(ops)))


def gen(N):
    out = []
    out.append("template <class C>")
    out.append("__device__ __forceinline__ void fp_mul_comba%d(Fp<C>& r, const Fp<C>& a, const Fp<C>& b) {" % N)
    out.append("  static_assert(C::N == %d, \"limb count\");" % N)
    out.append("  uint32_t m[%d], t[%d];" % (N, N))
    out.append("  uint64_t acc = 0, cy;  // cy: SGPR pair that carries the odd products' carry-outs")
    out.append("  uint32_t ext = 0;")
    for k in range(2 * N):
        out.append("  {  // column %d" % k)
        prods = []
        lo, hi = max(0, k - N + 1), min(k, N - 1)
        for i in range(lo, hi + 1):
            prods.append(("a.l[%d]" % i, "v", "b.l[%d]" % (k - i), "v"))
        for i in range(lo, hi + 1):
            if k < N and i == k:
                continue  # m[k] * P[0] comes after m[k] is known
            prods.append(("m[%d]" % i, "v", "(uint32_t)C::P[%d]" % (k - i), "s"))
        if k < 2 * N - 1:
            emit_stmt(out, prods)
        if k < N:
            out.append("    m[%d] = (uint32_t)acc * C::INV;" % k)
            emit_stmt(out, [("m[%d]" % k, "v", "(uint32_t)C::P[0]", "s")])
        elif k < 2 * N - 1:
            out.append("    t[%d] = (uint32_t)acc;" % (k - N))
        else:
            out.append("    t[%d] = (uint32_t)acc;" % (k - N))
        if k < 2 * N - 1:
            out.append("    acc = (acc >> 32) | ((uint64_t)ext << 32);")
            out.append("    ext = 0;")
        out.append("  }")
    out.append("  // a, b < p  =>  result < 2p < 2^(32N): one conditional subtraction")
    out.append("  fp_reduce_once<C>(r, t);")
    out.append("}")
    return "\n".join(out)


def gen_dual(N):
    """r = (a*b + c*d) * R^-1 mod p in one pass: both products share the column accumulator and ONE Montgomery
    reduction (2 N^2 + N^2 limb products instead of 4 N^2 for two separate multiplications).  Valid because
    a*b + c*d < 2 p^2 and 2p < R/4 for the supported moduli, so the reduced value is < 2p."""
    out = []
    out.append("template <class C>")
    out.append("__device__ __forceinline__ void fp_mul2_comba%d(Fp<C>& r, const Fp<C>& a, const Fp<C>& b, const Fp<C>& c, const Fp<C>& d) {" % N)
    out.append("  static_assert(C::N == %d, \"limb count\");" % N)
    out.append("  uint32_t m[%d], t[%d];" % (N, N))
    out.append("  uint64_t acc = 0, cy;  // cy: SGPR pair that carries the odd products' carry-outs")
    out.append("  uint32_t ext = 0;")
    for k in range(2 * N):
        out.append("  {  // column %d" % k)
        prods = []
        lo, hi = max(0, k - N + 1), min(k, N - 1)
        for i in range(lo, hi + 1):
            prods.append(("a.l[%d]" % i, "v", "b.l[%d]" % (k - i), "v"))
        for i in range(lo, hi + 1):
            prods.append(("c.l[%d]" % i, "v", "d.l[%d]" % (k - i), "v"))
        for i in range(lo, hi + 1):
            if k < N and i == k:
                continue
            prods.append(("m[%d]" % i, "v", "(uint32_t)C::P[%d]" % (k - i), "s"))
        if k < 2 * N - 1:
            emit_stmt(out, prods)
        if k < N:
            out.append("    m[%d] = (uint32_t)acc * C::INV;" % k)
            emit_stmt(out, [("m[%d]" % k, "v", "(uint32_t)C::P[0]", "s")])
        else:
            out.append("    t[%d] = (uint32_t)acc;" % (k - N))
        if k < 2 * N - 1:
            out.append("    acc = (acc >> 32) | ((uint64_t)ext << 32);")
            out.append("    ext = 0;")
        out.append("  }")
    out.append("  fp_reduce_once<C>(r, t);")
    out.append("}")
    return "\n".join(out)


print("// GENERATED by tools/gen_fp_comba.py -- do not edit.  Device-only (gfx950 inline asm).")
print(gen(8))
print()
print(gen(12))
print()
print(gen_dual(8))
print()
print(gen_dual(12))
