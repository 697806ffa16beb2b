#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ from oracle/pyref.py.

The reference (Go; gnark-crypto / kilic arithmetic not in its tree, no Go toolchain in this image)
cannot be run here, so these vectors come from the pure-Python big-integer restatement, whose
constants are pinned to the reference's known-answer values (tests/test_oracle_pinned.py).
Inputs are deterministic: SHA-256 counter DRBG, seed "mlhip-vec-1" (BASELINE.md section 3).

Run:  python tests/golden/gen_golden.py         (about a minute)
Files: <curve>.json          small cases, hex strings of the C-ABI byte layout (Montgomery, LE limbs)
       <curve>_msm1000.bin   n = 1000 G1 MSM: points | scalars (plain LE) | expected affine result
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref as R  # noqa: E402


def hx(b: bytes) -> str:
    return b.hex()


def msm_case(cp, name, pts, scs, group=1):
    if group == 1:
        exp = R.g1_msm(cp, pts, scs)
        return dict(
            name=name,
            points=[hx(R.g1_to_mont_bytes(cp, p)) for p in pts],
            scalars=[hx((s % (1 << 256)).to_bytes(32, "little")) for s in scs],
            scalars_int=[str(s) for s in scs],
            expected=hx(R.g1_to_mont_bytes(cp, exp)),
            expected_wire=hx(R.g1_wire_compressed(cp, exp)),
        )
    exp = R.g2_msm(cp, pts, scs)
    return dict(
        name=name,
        points=[hx(R.g2_to_mont_bytes(cp, p)) for p in pts],
        scalars=[hx((s % (1 << 256)).to_bytes(32, "little")) for s in scs],
        scalars_int=[str(s) for s in scs],
        expected=hx(R.g2_to_mont_bytes(cp, exp)),
    )


def gen_curve(cp):
    T = R.tower(cp)
    d = R.Drbg("golden/" + cp.name)
    out = dict(curve=cp.name, curve_id=cp.curve_id, p=hex(cp.p), r=hex(cp.r), fexp_cofactor=str(cp.fexp_cofactor))
    g2 = R.g2_generator(cp)
    out["g1_gen"] = hx(R.g1_to_mont_bytes(cp, cp.g1))
    out["g2_gen"] = hx(R.g2_to_mont_bytes(cp, g2))
    out["g2_gen_coords"] = [[str(g2[0][0]), str(g2[0][1])], [str(g2[1][0]), str(g2[1][1])]]

    # ---- field known-answer values (Montgomery multiply)
    fp = []
    for _ in range(8):
        a, b = d.below(cp.p), d.below(cp.p)
        fp.append(dict(a=hx(R.fp_to_mont_bytes(cp, a)), b=hx(R.fp_to_mont_bytes(cp, b)), ab=hx(R.fp_to_mont_bytes(cp, a * b % cp.p))))
    for a, b in ((0, 5), (1, cp.p - 1), (cp.p - 1, cp.p - 1)):
        fp.append(dict(a=hx(R.fp_to_mont_bytes(cp, a)), b=hx(R.fp_to_mont_bytes(cp, b)), ab=hx(R.fp_to_mont_bytes(cp, a * b % cp.p))))
    out["fp_mul"] = fp

    # ---- G1 MSM cases
    cases = []
    pts = [R.random_g1(cp, d) for _ in range(16)]
    rs = lambda: d.below(cp.r)  # noqa: E731
    cases.append(msm_case(cp, "n1", pts[:1], [rs()]))
    cases.append(msm_case(cp, "n2", pts[:2], [rs(), rs()]))
    cases.append(msm_case(cp, "n10_random", pts[:10], [rs() for _ in range(10)]))  # math_test.go:323-346 shape
    cases.append(msm_case(cp, "zero_scalars", pts[:4], [0, 0, 0, 0]))
    cases.append(msm_case(cp, "one_scalars", pts[:4], [1, 1, 1, 1]))
    cases.append(msm_case(cp, "r_minus_1", pts[:3], [cp.r - 1, cp.r - 1, 1]))
    cases.append(msm_case(cp, "infinity_points", [None, pts[0], None, pts[1]], [rs(), rs(), rs(), rs()]))
    cases.append(msm_case(cp, "duplicate_points", [pts[0]] * 6 + [pts[1]] * 3, [rs() for _ in range(9)]))
    s0 = rs()
    cases.append(msm_case(cp, "all_equal_scalars", pts[:12], [s0] * 12))
    cases.append(msm_case(cp, "p_and_minus_p", [pts[0], R.g1_neg(cp, pts[0]), pts[1]], [s0, s0, 7]))
    cases.append(msm_case(cp, "cancel_to_infinity", [pts[0], R.g1_neg(cp, pts[0])], [s0, s0]))
    cases.append(msm_case(cp, "small_scalars", pts[:8], [d.below(1 << 32) for _ in range(8)]))
    cases.append(msm_case(cp, "window_boundaries", pts[:6], [(1 << 15), (1 << 16) - 1, (1 << 16), (1 << 15) + 1, (1 << 255) % cp.r, cp.r - 2]))
    # scalars >= r and 'negative' (two's complement of a BaseZr: driver/common/big.go:101-113): plain encoding only
    cases.append(msm_case(cp, "unreduced_scalars", pts[:4], [cp.r, cp.r + 5, (1 << 256) - 1, 2 * cp.r + 3]))
    out["msm_g1"] = cases

    # ---- G2 MSM cases
    q = [R.random_g2(cp, d) for _ in range(6)]
    c2 = []
    c2.append(msm_case(cp, "n1", q[:1], [rs()], 2))
    c2.append(msm_case(cp, "n6_random", q, [rs() for _ in range(6)], 2))
    c2.append(msm_case(cp, "edge", [q[0], None, q[0], R.g2_neg(cp, q[1]), q[1]], [rs(), rs(), 0, s0, s0], 2))
    out["msm_g2"] = c2

    # ---- pairings
    pr = []
    pairs = [(cp.g1, g2), (pts[0], q[0]), (pts[1], q[1]), (pts[2], q[2])]
    for P, Q in pairs:
        pr.append(dict(g1=hx(R.g1_to_mont_bytes(cp, P)), g2=hx(R.g2_to_mont_bytes(cp, Q)), fexp=hx(R.gt_to_mont_bytes(cp, R.pairing(cp, P, Q)))))
    out["pairing"] = pr
    out["gen_gt_wire"] = hx(R.gt_wire_bytes(cp, R.pairing(cp, cp.g1, g2)))
    # Pairing2 (shared Miller loop) and bilinearity scalars
    f2 = R.final_exp(cp, R.miller_loop(cp, [(pts[0], q[0]), (pts[1], q[1])]))
    out["pairing2"] = dict(
        g1=[hx(R.g1_to_mont_bytes(cp, pts[0])), hx(R.g1_to_mont_bytes(cp, pts[1]))],
        g2=[hx(R.g2_to_mont_bytes(cp, q[0])), hx(R.g2_to_mont_bytes(cp, q[1]))],
        fexp=hx(R.gt_to_mont_bytes(cp, f2)),
    )
    a, b = rs(), rs()
    out["bilinear"] = dict(
        a=str(a),
        b=str(b),
        g1=hx(R.g1_to_mont_bytes(cp, R.g1_mul(cp, cp.g1, a))),
        g2=hx(R.g2_to_mont_bytes(cp, R.g2_mul(cp, g2, b))),
        fexp=hx(R.gt_to_mont_bytes(cp, T.f12_pow(R.pairing(cp, cp.g1, g2), a * b % cp.r))),
    )
    # a raw Miller-loop value and its final exponentiation (FExp input/output pair)
    ml = R.miller_loop(cp, [(pts[3], q[3])])
    out["fexp_io"] = dict(input=hx(R.gt_to_mont_bytes(cp, ml)), output=hx(R.gt_to_mont_bytes(cp, R.final_exp(cp, ml))))
    return out


def gen_msm1000(cp):
    d = R.Drbg("msm1000/" + cp.name)
    n = 1000
    pts = [R.random_g1(cp, d) for _ in range(n)]
    scs = [d.below(cp.r) for _ in range(n)]
    exp = R.g1_msm(cp, pts, scs)
    blob = b"".join(R.g1_to_mont_bytes(cp, p) for p in pts) + b"".join(s.to_bytes(32, "little") for s in scs) + R.g1_to_mont_bytes(cp, exp)
    return blob


def main():
    for cp in (R.BN254, R.BLS12_381, R.BLS12_377):
        tag = cp.name.lower().replace("-", "_")
        with open(os.path.join(HERE, tag + ".json"), "w") as f:
            json.dump(gen_curve(cp), f, indent=1)
        with open(os.path.join(HERE, tag + "_msm1000.bin"), "wb") as f:
            f.write(gen_msm1000(cp))
        print("wrote", tag, flush=True)


if __name__ == "__main__":
    main()
