#!/usr/bin/env python3
"""Generates mathlib_amd/csrc/fp28_comba.inc: the gfx950 device body of the carry-free 28-bit-limb Montgomery
product (fp28.h) for N28 = 10 and N28 = 14 limbs -- plain product, square and fused dual product.

Why asm at all (the portable C++ in fp28.h compiles to the same v_mad_i64_i32 products): hipcc re-associates each
column sum so that the carry of the previous column is added last, which costs one v_lshl_add_u64 and one
v_ashrrev_i64 per column -- both issue at the multiplier's quarter rate, like v_mad itself
(profiles/r01_ubench_int.txt) -- and makes the carry-free form no faster than the saturated one
(profiles/r01_ubench_madd.txt, V3).  Here every column is ONE accumulator chain that starts from the shifted
carry, and the 64-bit shift is a full-rate v_alignbit_b32 + v_ashrrev_i32.  No instruction reads a carry-out
(sdst = vcc, never consumed), so there is no VALU-writes-SGPR hazard to schedule around.

Run: python tools/gen_fp28_comba.py > mathlib_amd/csrc/fp28_comba.inc
"""

MAXOPS = 30


def emit_chain(out, prods):
    """prods: list of (x_expr, x_constraint, y_expr, y_constraint); acc += sum x*y, one v_mad_i64_i32 each."""
    per = (MAXOPS - 1) // 2
    for off in range(0, len(prods), per):
        chunk = prods[off : off + per]
        ops, lines = [], []
        for x, xc, y, yc in chunk:
            lines.append("v_mad_i64_i32 %%0, vcc, %%%d, %%%d, %%0" % (1 + len(ops), 2 + len(ops)))
            ops.append('"%s"(%s)' % (xc, x))
            ops.append('"%s"(%s)' % (yc, y))
        out.append('    asm("%s" : "+v"(acc) : %s : "vcc");' % ("\\n\\t".join(lines), ", ".join(ops)))


def emit_chain_from(out, dst, start, prods):
    """dst = start + sum x*y (start: '0' or a C expression of an int64 VGPR pair); dst is a fresh register pair, so the
    value in `start` survives."""
    per = (MAXOPS - 2) // 2
    assert len(prods) <= per
    ops, lines = [], []
    for n, (x, xc, y, yc) in enumerate(prods):
        if n == 0:
            c = "0" if start == "0" else "%%%d" % (1 + 2 * len(prods))
        else:
            c = "%0"
        lines.append("v_mad_i64_i32 %%0, vcc, %%%d, %%%d, %s" % (1 + len(ops), 2 + len(ops), c))
        ops.append('"%s"(%s)' % (xc, x))
        ops.append('"%s"(%s)' % (yc, y))
    if start != "0":
        ops.append('"v"(%s)' % start)
    out.append('    asm("%s" : "=&v"(%s) : %s : "vcc");' % ("\\n\\t".join(lines), dst, ", ".join(ops)))


def gen_k2(L):
    """Fp2 product (u^2 = -1) on ONE lane, Karatsuba over the components with the three limb products interleaved
    column by column: c0 = a0 b0 - a1 b1, c1 = (a0 + a1)(b0 + b1) - a0 b0 - a1 b1, two Montgomery reductions.  Per column:
    P0 = sum a0 b0 (fresh chain), S = P0 + sum a1 b1 (chain that starts from P0), c1 += sum s t + sum m1 p, c0 += sum m0 p
    (chains that start from the carries), then c1 -= S and c0 += 2 P0 - S in plain 64-bit arithmetic."""
    out = []
    out.append("template <class C>")
    out.append("__device__ __forceinline__ void fp28_k2mul_dev%d(Fp28<C>& r0, Fp28<C>& r1, const Fp28<C>& a0, const Fp28<C>& a1, const Fp28<C>& b0, const Fp28<C>& b1) {" % L)
    out.append('  static_assert(C::N28 == %d, "limb count");' % L)
    out.append("  int32_t s[%d], t[%d], m0[%d], m1[%d], t0[%d], t1[%d];" % (L, L, L, L, L, L))
    out.append("  for (int i = 0; i < %d; i++) { s[i] = a0.l[i] + a1.l[i]; t[i] = b0.l[i] + b1.l[i]; }" % L)
    out.append("  int64_t c0 = 0, c1 = 0;")
    for k in range(2 * L - 1):
        lo, hi = max(0, k - L + 1), min(k, L - 1)
        out.append("  {  // column %d" % k)
        out.append("    int64_t p0, sm;")
        emit_chain_from(out, "p0", "0", [("a0.l[%d]" % i, "v", "b0.l[%d]" % (k - i), "v") for i in range(lo, hi + 1)])
        emit_chain_from(out, "sm", "p0", [("a1.l[%d]" % i, "v", "b1.l[%d]" % (k - i), "v") for i in range(lo, hi + 1)])
        red = [i for i in range(lo, hi + 1) if not (k < L and i == k)]
        # c1 chain: s t products, then the reduction terms
        tmp = []
        emit_chain(tmp, [("s[%d]" % i, "v", "t[%d]" % (k - i), "v") for i in range(lo, hi + 1)] + [("m1[%d]" % i, "v", "C::P28[%d]" % (k - i), "s") for i in red])
        out.extend(x.replace('"+v"(acc)', '"+v"(c1)') for x in tmp)
        tmp = []
        emit_chain(tmp, [("m0[%d]" % i, "v", "C::P28[%d]" % (k - i), "s") for i in red])
        out.extend(x.replace('"+v"(acc)', '"+v"(c0)') for x in tmp)
        out.append("    c1 -= sm;")
        out.append("    c0 += 2 * p0 - sm;")
        if k < L:
            for c, m in (("c0", "m0"), ("c1", "m1")):
                out.append("    %s[%d] = (int32_t)(((uint32_t)%s * C::PINV28) & MASK28);" % (m, k, c))
                tmp = []
                emit_chain(tmp, [("%s[%d]" % (m, k), "v", "C::P28[0]", "s")])
                out.extend(x.replace('"+v"(acc)', '"+v"(%s)' % c) for x in tmp)
        else:
            out.append("    t0[%d] = (int32_t)((uint32_t)c0 & MASK28);" % (k - L))
            out.append("    t1[%d] = (int32_t)((uint32_t)c1 & MASK28);" % (k - L))
        out.append("    c0 = fp28_shift_dev(c0);")
        out.append("    c1 = fp28_shift_dev(c1);")
        out.append("  }")
    out.append("  t0[%d] = (int32_t)c0;" % (L - 1))
    out.append("  t1[%d] = (int32_t)c1;" % (L - 1))
    out.append("  for (int i = 0; i < %d; i++) { r0.l[i] = t0[i]; r1.l[i] = t1[i]; }" % L)
    out.append("}")
    return "\n".join(out)


def gen(L, kind):
    name = {"mul": "fp28_mul_dev%d", "sqr": "fp28_sqr_dev%d", "mul2": "fp28_mul2_dev%d"}[kind] % L
    args = {"mul": "const Fp28<C>& a, const Fp28<C>& b", "sqr": "const Fp28<C>& a",
            "mul2": "const Fp28<C>& a, const Fp28<C>& b, const Fp28<C>& c, const Fp28<C>& d"}[kind]
    out = []
    out.append("template <class C>")
    out.append("__device__ __forceinline__ void %s(Fp28<C>& r, %s) {" % (name, args))
    out.append('  static_assert(C::N28 == %d, "limb count");' % L)
    out.append("  int32_t m[%d], t[%d];" % (L, L))
    if kind == "sqr":
        out.append("  int32_t a2[%d];" % L)
        out.append("  for (int i = 0; i < %d; i++) a2[i] = a.l[i] + a.l[i];" % L)
    out.append("  int64_t acc = 0;")
    for k in range(2 * L - 1):
        lo, hi = max(0, k - L + 1), min(k, L - 1)
        prods = []
        for i in range(lo, hi + 1):
            j = k - i
            if kind == "sqr":
                if i < j:
                    prods.append(("a.l[%d]" % i, "v", "a2[%d]" % j, "v"))
                elif i == j:
                    prods.append(("a.l[%d]" % i, "v", "a.l[%d]" % i, "v"))
            else:
                prods.append(("a.l[%d]" % i, "v", "b.l[%d]" % j, "v"))
        if kind == "mul2":
            for i in range(lo, hi + 1):
                prods.append(("c.l[%d]" % i, "v", "d.l[%d]" % (k - i), "v"))
        for i in range(lo, hi + 1):
            if k < L and i == k:
                continue
            prods.append(("m[%d]" % i, "v", "C::P28[%d]" % (k - i), "s"))
        out.append("  {  // column %d" % k)
        emit_chain(out, prods)
        if k < L:
            out.append("    m[%d] = (int32_t)(((uint32_t)acc * C::PINV28) & MASK28);" % k)
            emit_chain(out, [("m[%d]" % k, "v", "C::P28[0]", "s")])
        else:
            out.append("    t[%d] = (int32_t)((uint32_t)acc & MASK28);" % (k - L))
        out.append("    acc = fp28_shift_dev(acc);")
        out.append("  }")
    out.append("  t[%d] = (int32_t)acc;" % (L - 1))
    out.append("  for (int i = 0; i < %d; i++) r.l[i] = t[i];" % L)
    out.append("}")
    return "\n".join(out)


print("// GENERATED by tools/gen_fp28_comba.py -- do not edit.  Device-only (gfx950 inline asm).")
print("""// acc >> 28 (arithmetic): one v_ashrrev_i64.  It issues at the multiplier's rate, but measured against the two
// full-rate instructions it replaces (v_alignbit_b32 + v_ashrrev_i32, still selectable with MLHIP_FP28_SHIFT32) it is the
// cheaper form: k_accumulate28 2.380 -> 2.350 ms, 65 536 pairings 19.05 -> 18.75 ms on one box (profiles/r02_ab_shift64.txt)
__device__ __forceinline__ int64_t fp28_shift_dev(int64_t acc) {
#if !defined(MLHIP_FP28_SHIFT32)
  return acc >> 28;  // hipcc selects v_ashrrev_i64; plain C keeps the statement out of the asm-boundary padding
#else
  const uint32_t lo = (uint32_t)acc;
  const int32_t hi = (int32_t)(acc >> 32);
  uint32_t nlo;
  asm("v_alignbit_b32 %0, %1, %2, 28" : "=v"(nlo) : "v"(hi), "v"(lo));
  return (int64_t)(((uint64_t)(uint32_t)(hi >> 28) << 32) | nlo);
#endif
}
""")
for L in (10, 14):
    for kind in ("mul", "sqr", "mul2"):
        print(gen(L, kind))
        print()
    print(gen_k2(L))
    print()
