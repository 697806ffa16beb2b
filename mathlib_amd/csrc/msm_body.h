// msm_body.h -- per-thread bodies of the Pippenger MSM kernels (gfx950), shared with the host-side
// emulation used by tests/test_host_math.py (each body is __host__ __device__ and takes the global
// thread index explicitly, so the CPU test can replay a launch thread by thread).
//
// Replaces gnark-crypto's MultiExp behind the reference's MultiScalarMul
// (driver/gurvy/bls12381/bls12-381.go:766-783, driver/gurvy/bn254.go:232-245,
// driver/gurvy/bls12-377.go:229-242): signed-digit windows, bucket accumulation in XYZZ
// coordinates, bucket reduction, window combination.  Data layout and pipeline: DESIGN.md section 3.
#pragma once
#include <cstring>

#include "ec.h"

namespace mlhip {

// ---- scalar handling -------------------------------------------------------------------------
// Scalars arrive as 8 little-endian 32-bit words: either gnark's fr.Element (Montgomery, R = 2^256;
// what driver/gurvy/bls12381 passes, bls12-381.go:772) or a plain integer (any 256-bit value; it is
// reduced mod r here, as fr.Element.SetBigInt does for the BaseZr curves, bn254.go:239).
template <class C>
MLHIP_HD void fr_canonical(uint32_t (&s)[8], const uint32_t* in, bool mont) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = in[i];
  t[8] = 0;
  if (mont) {
    // Montgomery reduction: t <- t * 2^-256 mod r
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t m = t[0] * C::FR_INV;
      uint64_t acc = (uint64_t)m * C::FR[0] + t[0];
      uint64_t c = acc >> 32;
#pragma unroll
      for (int j = 1; j < 8; j++) {
        acc = (uint64_t)m * C::FR[j] + t[j] + c;
        t[j - 1] = (uint32_t)acc;
        c = acc >> 32;
      }
      acc = (uint64_t)t[8] + c;
      t[7] = (uint32_t)acc;
      t[8] = (uint32_t)(acc >> 32);
    }
  }
  // reduce below r (a 256-bit input is < 16 r for every supported curve)
  for (int it = 0; it < 16; it++) {
    uint32_t d[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t x = (uint64_t)t[i] - C::FR[i] - br;
      d[i] = (uint32_t)x;
      br = (x >> 32) & 1;
    }
    bool ge = (t[8] != 0) | (br == 0);
    if (!ge) break;
    t[8] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = d[i];
  }
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = t[i];
}

MLHIP_HD int msm_num_windows(int fr_bits, int c) { return (fr_bits + 1 + c - 1) / c; }

// Window layout.  The FR_BITS + 1 bits of a scalar (the extra bit absorbs the last carry) are spread over
// W = ceil((FR_BITS + 1) / c) windows as evenly as possible: the first `rem` windows are base + 1 bits wide, the others
// base bits, base = (FR_BITS + 1) / W.  For 256 = 16 x 16 bits (BLS12-381 at c = 16) every window is c bits; where c
// does not divide -- BLS12-377 and BN254 at c = 16 (254 / 255 bits), every curve at other widths -- this replaces a full
// set of c-bit windows plus a sparse top window, whose few buckets were several times longer than all the others (one
// thread per bucket: the accumulation waited for them), by windows that differ by one bit.
struct WinLayout {
  int W, base, rem;
};
MLHIP_HD WinLayout msm_win_layout(int fr_bits, int c) {
  WinLayout l;
  l.W = msm_num_windows(fr_bits, c);
  l.base = (fr_bits + 1) / l.W;
  l.rem = (fr_bits + 1) % l.W;
  return l;
}
MLHIP_HD int msm_win_off(int wbase, int wrem, int w) { return w * wbase + (w < wrem ? w : wrem); }
MLHIP_HD int msm_win_bits(int wbase, int wrem, int w) { return wbase + (w < wrem ? 1 : 0); }

// Signed digit of window w: v = bits [off, off + width) + carry; v > 2^(width-1) => v -= 2^width, carry 1.
// Returns the magnitude (0 = skip; bucket = magnitude - 1), sets neg and the carry into the next window.
MLHIP_HD uint32_t msm_window_digit(const uint32_t (&s)[8], int off, int width, uint32_t& carry, uint32_t& neg) {
  uint32_t v = 0;
  if (off < 256) {
    const int word = off >> 5, sh = off & 31;
    uint64_t two = s[word];
    if (word + 1 < 8) two |= (uint64_t)s[word + 1] << 32;
    v = (uint32_t)((two >> sh) & ((1u << width) - 1));
  }
  v += carry;
  const uint32_t half = 1u << (width - 1);
  if (v > half) {
    // v == 2^width (all-ones window plus carry) is digit 0 with a carry, not "minus zero"
    carry = 1;
    neg = 1;
    return (1u << width) - v;
  }
  carry = 0;
  neg = 0;
  return v;
}

// Encoded as 0 (skip) or (magnitude << 1) | sign with magnitude in [1, 2^(width-1)] -> bucket magnitude-1.
template <class C>
MLHIP_HD void msm_digits_body(size_t i, size_t n, const uint32_t* scalars, bool mont, int c, int W,
                              uint32_t* digits /* [W][n] */) {
  uint32_t s[8];
  fr_canonical<C>(s, scalars + 8 * i, mont);
  const WinLayout l = msm_win_layout(C::FR_BITS, c);
  uint32_t carry = 0, neg = 0;
  for (int w = 0; w < W; w++) {
    const uint32_t mag = msm_window_digit(s, msm_win_off(l.base, l.rem, w), msm_win_bits(l.base, l.rem, w), carry, neg);
    digits[(size_t)w * n + i] = mag ? ((mag << 1) | neg) : 0u;
  }
}

// ---- bucket accumulation ---------------------------------------------------------------------
// One thread owns bucket g = w*M + b: adds the points listed in sorted[offset .. offset+count).
template <class F>
MLHIP_HD void msm_accumulate_range(XYZZ<F>& acc, const Affine<F>* points, const uint32_t* sorted, size_t begin,
                                   size_t end, size_t stride) {
  if (begin >= end) return;
  // software prefetch: the next entry's index and point are loaded before the current mixed addition,
  // so the two dependent gathers (index, then a random 96/192-byte row) overlap ~30k cycles of arithmetic
  uint32_t e = sorted[begin];
  Affine<F> p = points[e & 0x7fffffffu];
  for (size_t k = begin; k < end; k += stride) {
    const size_t kn = k + stride;
    uint32_t en = e;
    Affine<F> pn = p;
    if (kn < end) {
      en = sorted[kn];
      pn = points[en & 0x7fffffffu];
    }
    xyzz_madd<F>(acc, p, (e >> 31) != 0);
    e = en;
    p = pn;
  }
}

// ---- bucket reduction, level 1 ---------------------------------------------------------------
// Thread t of window w owns buckets [t*L, (t+1)*L): A = sum B_b ; W0 = sum_i i * B_{tL+i}.
// `add(acc, q)` is the group addition to use: the kernels pass an out-of-line copy (one ~90 KB body per
// kernel instead of three inlined ones), the host test passes xyzz_add itself.
template <class F, class AddFn>
MLHIP_HD void msm_chunk_body(size_t g, const XYZZ<F>* buckets, XYZZ<F>* A, XYZZ<F>* W0, int l_eff, AddFn add) {
  const XYZZ<F>* b = buckets + g * (size_t)l_eff;
  XYZZ<F> acc, w0;
  xyzz_set_inf<F>(acc);
  xyzz_set_inf<F>(w0);
  for (int i = l_eff - 1; i >= 1; i--) {
    add(acc, b[i]);
    add(w0, acc);
  }
  add(acc, b[0]);
  A[g] = acc;
  W0[g] = w0;
}

// ---- host tail of one window (msm_plan.h: host_tail; here so that the host-math test library can run it) ------------
// V = out[0..3] summed + 2^lgL sum_k 2^k out[4 + k]: nb + lgL doublings and nb + 4 additions
struct HostTailHeader {
  int nb, lgL, nsel, pad;
};
template <class F>
void host_tail_window(const void* in, int w, void* out) {
  const HostTailHeader& h = *static_cast<const HostTailHeader*>(in);
  const XYZZ<F>* o = reinterpret_cast<const XYZZ<F>*>(static_cast<const unsigned char*>(in) + sizeof(HostTailHeader)) + (size_t)w * h.nsel;
  XYZZ<F> acc, d;
  xyzz_set_inf<F>(acc);
  for (int k = h.nb - 1; k >= 0; k--) {
    xyzz_dbl<F>(d, acc);
    acc = d;
    xyzz_add<F>(acc, o[4 + k]);
  }
  for (int k = 0; k < h.lgL; k++) {
    xyzz_dbl<F>(d, acc);
    acc = d;
  }
  for (int q = 0; q < 4; q++) xyzz_add<F>(acc, o[q]);
  memcpy(out, &acc, sizeof(acc));
}

}  // namespace mlhip
