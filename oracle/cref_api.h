/* oracle/cref_api.h -- TEST INFRASTRUCTURE (oracle): per-curve entry points behind cref.c's exported
 * functions (byte buffers in the C-ABI layout of include/mlhip.h: Montgomery, little-endian limbs). */

/* scalar: 4 LE words, Montgomery (fr.Element) or plain (any 256-bit value) -> canonical < r
 * (fr.Element.SetBigInt semantics of driver/gurvy/bn254.go:239; direct fr.Element copy of bls12-381.go:772) */
static void FN(fr_canonical)(uint64_t s[4], const uint64_t in[4], int mont) {
  uint64_t t[5];
  for (int i = 0; i < 4; i++) t[i] = in[i];
  t[4] = 0;
  if (mont) {
    for (int i = 0; i < 4; i++) {
      uint64_t m = t[0] * K(_FR_INV);
      unsigned __int128 acc = (unsigned __int128)m * K(_FR)[0] + t[0];
      unsigned __int128 c = acc >> 64;
      for (int j = 1; j < 4; j++) {
        acc = (unsigned __int128)m * K(_FR)[j] + t[j] + c;
        t[j - 1] = (uint64_t)acc;
        c = acc >> 64;
      }
      acc = (unsigned __int128)t[4] + c;
      t[3] = (uint64_t)acc;
      t[4] = (uint64_t)(acc >> 64);
    }
  }
  for (;;) {
    uint64_t d[4];
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 x = (unsigned __int128)t[i] - K(_FR)[i] - br;
      d[i] = (uint64_t)x;
      br = (x >> 64) & 1;
    }
    if (!(t[4] != 0 || br == 0)) break;
    t[4] = 0;
    for (int i = 0; i < 4; i++) t[i] = d[i];
  }
  for (int i = 0; i < 4; i++) s[i] = t[i];
}

static uint64_t* FN(canon_scalars)(const void* sc, int mont, size_t n) {
  uint64_t* out = (uint64_t*)malloc((n ? n : 1) * 32);
  for (size_t i = 0; i < n; i++) {
    uint64_t in[4];
    memcpy(in, (const char*)sc + 32 * i, 32);
    FN(fr_canonical)(out + 4 * i, in, mont);
  }
  return out;
}

static int FN(cref_fp_mul)(const void* a, const void* b, void* out) {
  FN(fp) x, y, r;
  memcpy(&x, a, sizeof(x));
  memcpy(&y, b, sizeof(y));
  FN(fp_mul)(&r, &x, &y);
  memcpy(out, &r, sizeof(r));
  return 0;
}

static int FN(cref_msm_g1)(const void* pts, const void* sc, int mont, size_t n, int c, int threads, void* out) {
  uint64_t* cs = FN(canon_scalars)(sc, mont, n);
  CR_CAT(aff_g1, CR_SUF) r;
  CR_CAT(msm_g1, CR_SUF)((const CR_CAT(aff_g1, CR_SUF)*)pts, cs, n, c, threads, &r);
  memcpy(out, &r, sizeof(r));
  free(cs);
  return 0;
}

static int FN(cref_msm_g2)(const void* pts, const void* sc, int mont, size_t n, int c, int threads, void* out) {
  uint64_t* cs = FN(canon_scalars)(sc, mont, n);
  CR_CAT(aff_g2, CR_SUF) r;
  CR_CAT(msm_g2, CR_SUF)((const CR_CAT(aff_g2, CR_SUF)*)pts, cs, n, c, threads, &r);
  memcpy(out, &r, sizeof(r));
  free(cs);
  return 0;
}

static int FN(cref_g1_mul)(const void* p, const void* k, int mont, void* out) {
  uint64_t in[4], s[4];
  memcpy(in, k, 32);
  FN(fr_canonical)(s, in, mont);
  CR_CAT(jac_g1, CR_SUF) j;
  CR_CAT(aff_g1, CR_SUF) a, r;
  memcpy(&a, p, sizeof(a));
  CR_CAT(scalar_mul_g1, CR_SUF)(&j, &a, s);
  CR_CAT(jac_to_aff_g1, CR_SUF)(&r, &j);
  memcpy(out, &r, sizeof(r));
  return 0;
}

static int FN(cref_g2_mul)(const void* p, const void* k, int mont, void* out) {
  uint64_t in[4], s[4];
  memcpy(in, k, 32);
  FN(fr_canonical)(s, in, mont);
  CR_CAT(jac_g2, CR_SUF) j;
  CR_CAT(aff_g2, CR_SUF) a, r;
  memcpy(&a, p, sizeof(a));
  CR_CAT(scalar_mul_g2, CR_SUF)(&j, &a, s);
  CR_CAT(jac_to_aff_g2, CR_SUF)(&r, &j);
  memcpy(out, &r, sizeof(r));
  return 0;
}

static int FN(cref_gen_g1)(const void* k0, const void* k1, size_t n, void* out) {
  uint64_t a[4], b[4], in[4];
  memcpy(in, k0, 32); FN(fr_canonical)(a, in, 0);
  memcpy(in, k1, 32); FN(fr_canonical)(b, in, 0);
  CR_CAT(aff_g1, CR_SUF) g;
  memcpy(&g, K(_G1), sizeof(g));
  CR_CAT(gen_points_g1, CR_SUF)(&g, a, b, n, (CR_CAT(aff_g1, CR_SUF)*)out);
  return 0;
}

static int FN(cref_gen_g2)(const void* k0, const void* k1, size_t n, void* out) {
  uint64_t a[4], b[4], in[4];
  memcpy(in, k0, 32); FN(fr_canonical)(a, in, 0);
  memcpy(in, k1, 32); FN(fr_canonical)(b, in, 0);
  CR_CAT(aff_g2, CR_SUF) g;
  memcpy(&g, K(_G2), sizeof(g));
  CR_CAT(gen_points_g2, CR_SUF)(&g, a, b, n, (CR_CAT(aff_g2, CR_SUF)*)out);
  return 0;
}

/* ---- batched pairing work split over pthreads */
typedef struct {
  int what; /* 0 miller, 1 final exp, 2 both */
  const FN(g1a)* g1;
  const FN(g2a)* g2;
  const FN(fp12)* in;
  FN(fp12)* out;
  size_t ppp, n;
  volatile long next;
} FN(pjob);

static void* FN(pair_worker)(void* arg) {
  FN(pjob)* j = (FN(pjob)*)arg;
  for (;;) {
    long i = __sync_fetch_and_add(&j->next, 1);
    if ((size_t)i >= j->n) break;
    FN(fp12) f, r;
    if (j->what == 1) {
      FN(final_exp)(&r, &j->in[i]);
      j->out[i] = r;
    } else {
      FN(miller_loop)(&f, j->g1 + i * j->ppp, j->g2 + i * j->ppp, (int)j->ppp);
      if (j->what == 2) { FN(final_exp)(&r, &f); j->out[i] = r; } else j->out[i] = f;
    }
  }
  return NULL;
}

static int FN(pair_run)(FN(pjob)* j, int threads) {
  j->next = 0;
  if (threads <= 1) { FN(pair_worker)(j); return 0; }
  pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
  for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, FN(pair_worker), j);
  for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
  free(th);
  return 0;
}

static int FN(cref_miller_loop)(const void* g1, const void* g2, size_t ppp, size_t n, void* out, int threads) {
  FN(pjob) j = {0, (const FN(g1a)*)g1, (const FN(g2a)*)g2, NULL, (FN(fp12)*)out, ppp, n, 0};
  return FN(pair_run)(&j, threads);
}
static int FN(cref_final_exp)(const void* in, size_t n, void* out, int threads) {
  FN(pjob) j = {1, NULL, NULL, (const FN(fp12)*)in, (FN(fp12)*)out, 1, n, 0};
  return FN(pair_run)(&j, threads);
}
static int FN(cref_pairing_batch)(const void* g1, const void* g2, size_t n, void* out, int threads) {
  FN(pjob) j = {2, (const FN(g1a)*)g1, (const FN(g2a)*)g2, NULL, (FN(fp12)*)out, 1, n, 0};
  return FN(pair_run)(&j, threads);
}
static int FN(cref_gt_mul)(const void* a, const void* b, size_t n, void* out) {
  for (size_t i = 0; i < n; i++) {
    FN(fp12) x, y, r;
    memcpy(&x, (const char*)a + i * sizeof(x), sizeof(x));
    memcpy(&y, (const char*)b + i * sizeof(y), sizeof(y));
    FN(fp12_mul)(&r, &x, &y);
    memcpy((char*)out + i * sizeof(r), &r, sizeof(r));
  }
  return 0;
}
