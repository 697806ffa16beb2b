/* oracle/cref.c -- TEST INFRASTRUCTURE (oracle), never part of the product path.
 *
 * Plain-C CPU restatement of the reference's hot path for BN254, BLS12-381 and BLS12-377:
 * G1/G2 multi-scalar multiplication (Pippenger, pthreads) and the optimal-ate pairing
 * (Miller loop + final exponentiation).  Used by tests/ as the fast checker for sizes the Python
 * oracle cannot reach, by bench.py as the `cpu_baseline` ("port"), and to generate synthetic inputs.
 * It is validated against oracle/pyref.py and tests/golden (tests/test_cref.py).
 *
 * Reference semantics restated (paths relative to /root/reference):
 *   MultiScalarMul   driver/gurvy/bls12381/bls12-381.go:766-783, driver/gurvy/bn254.go:232-245,
 *                    driver/gurvy/bls12-377.go:229-242, driver/kilic/bls12-381.go:247-254
 *   Pairing/Pairing2 driver/gurvy/bls12381/bls12-381.go:448-464 (Miller loop only)
 *   FExp             driver/gurvy/bls12381/bls12-381.go:466-468
 *   Fp multiply      driver/kilic/custom_generic.go:57-175
 * The arithmetic libraries themselves (gnark-crypto v0.20.1, kilic/bls12-381 v0.1.0: go.mod:6,15) are
 * not in the reference tree and there is no Go toolchain here; for MSM / pairing OUTPUTS the reference
 * holds no golden vector => parity unpinned by the reference (see oracle/pyref.py header).
 *
 * Build: make -C oracle   (gcc -O2, -lpthread) -> oracle/libcref.so
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cref_constants.h"

#define CR_CAT_(a, b) a##b
#define CR_CAT(a, b) CR_CAT_(a, b)
#define K(name) CR_CAT(CR_TAG, name)
#define FN(name) CR_CAT(name, CR_SUF)

/* BN254 */
#define CR_TAG BN254
#define CR_SUF _bn254
#define CR_FR_BITS BN254_FR_BITS
#include "cref_field.h"
#define EF(name) CR_CAT(CR_CAT(name, _g1), CR_SUF)
#define ET FN(fp)
#define E_add FN(fp_add)
#define E_sub FN(fp_sub)
#define E_mul FN(fp_mul)
#define E_sqr FN(fp_sqr)
#define E_dbl FN(fp_dbl)
#define E_neg FN(fp_neg)
#define E_inv FN(fp_inv)
#define E_is_zero FN(fp_is_zero)
#define E_eq FN(fp_eq)
#define E_one FN(fp_one)
#define E_zero FN(fp_zero)
#include "cref_ec.h"
#include "cref_undef_e.h"
#define EF(name) CR_CAT(CR_CAT(name, _g2), CR_SUF)
#define ET FN(fp2)
#define E_add FN(fp2_add)
#define E_sub FN(fp2_sub)
#define E_mul FN(fp2_mul)
#define E_sqr FN(fp2_sqr)
#define E_dbl FN(fp2_dbl)
#define E_neg FN(fp2_neg)
#define E_inv FN(fp2_inv)
#define E_is_zero FN(fp2_is_zero)
#define E_eq FN(fp2_eq)
#define E_one FN(fp2_one)
#define E_zero FN(fp2_zero)
#include "cref_ec.h"
#include "cref_undef_e.h"
#include "cref_api.h"
#undef NL
#undef CR_TAG
#undef CR_SUF
#undef CR_FR_BITS

/* BLS12-381 */
#define CR_TAG BLS381
#define CR_SUF _bls381
#define CR_FR_BITS BLS381_FR_BITS
#include "cref_field.h"
#define EF(name) CR_CAT(CR_CAT(name, _g1), CR_SUF)
#define ET FN(fp)
#define E_add FN(fp_add)
#define E_sub FN(fp_sub)
#define E_mul FN(fp_mul)
#define E_sqr FN(fp_sqr)
#define E_dbl FN(fp_dbl)
#define E_neg FN(fp_neg)
#define E_inv FN(fp_inv)
#define E_is_zero FN(fp_is_zero)
#define E_eq FN(fp_eq)
#define E_one FN(fp_one)
#define E_zero FN(fp_zero)
#include "cref_ec.h"
#include "cref_undef_e.h"
#define EF(name) CR_CAT(CR_CAT(name, _g2), CR_SUF)
#define ET FN(fp2)
#define E_add FN(fp2_add)
#define E_sub FN(fp2_sub)
#define E_mul FN(fp2_mul)
#define E_sqr FN(fp2_sqr)
#define E_dbl FN(fp2_dbl)
#define E_neg FN(fp2_neg)
#define E_inv FN(fp2_inv)
#define E_is_zero FN(fp2_is_zero)
#define E_eq FN(fp2_eq)
#define E_one FN(fp2_one)
#define E_zero FN(fp2_zero)
#include "cref_ec.h"
#include "cref_undef_e.h"
#include "cref_api.h"
#undef NL
#undef CR_TAG
#undef CR_SUF
#undef CR_FR_BITS

/* BLS12-377 */
#define CR_TAG BLS377
#define CR_SUF _bls377
#define CR_FR_BITS BLS377_FR_BITS
#include "cref_field.h"
#define EF(name) CR_CAT(CR_CAT(name, _g1), CR_SUF)
#define ET FN(fp)
#define E_add FN(fp_add)
#define E_sub FN(fp_sub)
#define E_mul FN(fp_mul)
#define E_sqr FN(fp_sqr)
#define E_dbl FN(fp_dbl)
#define E_neg FN(fp_neg)
#define E_inv FN(fp_inv)
#define E_is_zero FN(fp_is_zero)
#define E_eq FN(fp_eq)
#define E_one FN(fp_one)
#define E_zero FN(fp_zero)
#include "cref_ec.h"
#include "cref_undef_e.h"
#define EF(name) CR_CAT(CR_CAT(name, _g2), CR_SUF)
#define ET FN(fp2)
#define E_add FN(fp2_add)
#define E_sub FN(fp2_sub)
#define E_mul FN(fp2_mul)
#define E_sqr FN(fp2_sqr)
#define E_dbl FN(fp2_dbl)
#define E_neg FN(fp2_neg)
#define E_inv FN(fp2_inv)
#define E_is_zero FN(fp2_is_zero)
#define E_eq FN(fp2_eq)
#define E_one FN(fp2_one)
#define E_zero FN(fp2_zero)
#include "cref_ec.h"
#include "cref_undef_e.h"
#include "cref_api.h"
#undef NL
#undef CR_TAG
#undef CR_SUF
#undef CR_FR_BITS

/* ------------------------------------------------------------------ exported API (curve id dispatch) */

#define API3(name, proto, args)                                   \
  int name proto {                                                \
    switch (curve) {                                              \
      case 0: return CR_CAT(name, _bn254) args;                   \
      case 1: return CR_CAT(name, _bls381) args;                  \
      case 2: return CR_CAT(name, _bls377) args;                  \
      default: return -1;                                         \
    }                                                             \
  }

API3(cref_fp_mul, (int curve, const void* a, const void* b, void* out), (a, b, out))
API3(cref_msm_g1, (int curve, const void* pts, const void* sc, int mont, size_t n, int c, int threads, void* out), (pts, sc, mont, n, c, threads, out))
API3(cref_msm_g2, (int curve, const void* pts, const void* sc, int mont, size_t n, int c, int threads, void* out), (pts, sc, mont, n, c, threads, out))
API3(cref_miller_loop, (int curve, const void* g1, const void* g2, size_t ppp, size_t n, void* out, int threads), (g1, g2, ppp, n, out, threads))
API3(cref_final_exp, (int curve, const void* in, size_t n, void* out, int threads), (in, n, out, threads))
API3(cref_pairing_batch, (int curve, const void* g1, const void* g2, size_t n, void* out, int threads), (g1, g2, n, out, threads))
API3(cref_gt_mul, (int curve, const void* a, const void* b, size_t n, void* out), (a, b, n, out))
API3(cref_gen_g1, (int curve, const void* k0, const void* k1, size_t n, void* out), (k0, k1, n, out))
API3(cref_gen_g2, (int curve, const void* k0, const void* k1, size_t n, void* out), (k0, k1, n, out))
API3(cref_g1_mul, (int curve, const void* p, const void* k, int mont, void* out), (p, k, mont, out))
API3(cref_g2_mul, (int curve, const void* p, const void* k, int mont, void* out), (p, k, mont, out))
