// msm_fold_body.h -- per-thread bodies of the shifted-base-table kernels (msm_fold.h, msm_reduce.h: k_group_combine_q),
// shared with the host-side emulation of tests/test_host_math.py (hm_fold_msm), like msm_body.h.
#pragma once
#include "ec_jac.h"
#include "msm_body.h"

namespace mlhip {

// rows[j stride] = 2^off(j) P, j = 0 .. W - 1, off(j) = the first bit of digit j (msm_win_layout: the digits of a scalar
// are as wide as each other up to one bit, so no digit position is sparse): off(j) - off(j - 1) Jacobian doublings
// (ec_jac.h) per row and one inversion to make the row affine.  A base at infinity (0, 0), or one whose multiple reaches
// infinity (points of small order exist outside the prime-order subgroup), gives rows (0, 0): the accumulation kernels
// skip them.
template <class F>
MLHIP_HD void fold_rows_body(const Affine<F>& P, int fr_bits, int c, size_t stride, Affine<F>* rows) {
  const WinLayout wl = msm_win_layout(fr_bits, c);
  rows[0] = P;
  Jac<F> acc;
  if (F::is_zero(P.x) && F::is_zero(P.y)) {
    jac_set_inf<F>(acc);
  } else {
    acc.x = P.x;
    acc.y = P.y;
    F::one(acc.z);
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int j = 1; j < wl.W; j++) {
    const int steps = msm_win_off(wl.base, wl.rem, j) - msm_win_off(wl.base, wl.rem, j - 1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int k = 0; k < steps; k++) jac_dbl<F>(acc, acc);
    Affine<F> r;
    if (jac_is_inf<F>(acc)) {
      F::zero(r.x);
      F::zero(r.y);
    } else {
      typename F::T zi, z2;
      F::inv(zi, acc.z);
      F::sqr(z2, zi);
      F::mul(r.x, acc.x, z2);
      F::mul(z2, z2, zi);
      F::mul(r.y, acc.y, z2);
    }
    rows[(size_t)j * stride] = r;
  }
}

// k_group_combine_q: which of group g's nsel sums (two halves of sum W0, two halves of sum A, nb bit-masked sums of A) goes
// into output o of the combined window -- slot (g, h), h = 0 / 1 -- or -1 for none.  Outputs: 0 = sum W0, 1 = sum A, 2 and 3
// empty, 4 + k = B_k, the sum of the A[t'] with bit k of the global chunk index t' = g T + t set: bits below nb are bits of t
// (group g's own masked sum, slot h = 0), bits from nb on are bits of g (group g's plain sum, both halves).
MLHIP_HD int fold_combine_src(int o, int g, int h, int nb) {
  if (o == 0) return h;
  if (o == 1) return 2 + h;
  if (o >= 4 && o < 4 + nb) return h == 0 ? o : -1;
  if (o >= 4 + nb) return ((g >> (o - 4 - nb)) & 1) ? 2 + h : -1;
  return -1;
}

}  // namespace mlhip
