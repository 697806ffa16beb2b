/* oracle/cref_field.h -- TEST INFRASTRUCTURE (oracle): per-curve field towers and pairing, plain C.
 * Included once per curve by cref.c with CR_TAG (constant prefix) and CR_SUF (symbol suffix) defined.
 * CPU restatement (64-bit limbs, unsigned __int128) of the arithmetic behind the reference's hot path;
 * it is validated against oracle/pyref.py and the committed golden vectors (tests/test_cref.py), and is
 * the "port" CPU baseline of bench.py.  Never linked into libmlhip.so.
 *
 * Follows: Fp Montgomery multiply driver/kilic/custom_generic.go:57-175 (6x64 CIOS, final conditional
 * subtraction :166-174); Pairing/FExp semantics driver/gurvy/bls12381/bls12-381.go:448-468,
 * driver/gurvy/bn254.go:247-267, driver/gurvy/bls12-377.go:244-264.
 */
#define NL K(_NL)

typedef struct { uint64_t l[NL]; } FN(fp);
typedef struct { FN(fp) c0, c1; } FN(fp2);
typedef struct { FN(fp2) c0, c1, c2; } FN(fp6);
typedef struct { FN(fp6) c0, c1; } FN(fp12);


/* ---------------------------------------------------------------- Fp */
static inline void FN(fp_zero)(FN(fp)* r) { memset(r, 0, sizeof(*r)); }
static inline void FN(fp_one)(FN(fp)* r) { memcpy(r->l, K(_ONE), sizeof(r->l)); }
static inline int FN(fp_is_zero)(const FN(fp)* a) {
  uint64_t o = 0;
  for (int i = 0; i < NL; i++) o |= a->l[i];
  return o == 0;
}
static inline int FN(fp_eq)(const FN(fp)* a, const FN(fp)* b) { return memcmp(a, b, sizeof(*a)) == 0; }

static inline void FN(fp_add)(FN(fp)* r, const FN(fp)* a, const FN(fp)* b) {
  uint64_t t[NL], d[NL];
  unsigned __int128 c = 0;
  for (int i = 0; i < NL; i++) {
    c += (unsigned __int128)a->l[i] + b->l[i];
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  unsigned __int128 br = 0;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 s = (unsigned __int128)t[i] - K(_P)[i] - br;
    d[i] = (uint64_t)s;
    br = (s >> 64) & 1;
  }
  int ge = (c != 0) | (br == 0);
  for (int i = 0; i < NL; i++) r->l[i] = ge ? d[i] : t[i];
}

static inline void FN(fp_sub)(FN(fp)* r, const FN(fp)* a, const FN(fp)* b) {
  uint64_t d[NL];
  unsigned __int128 br = 0;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 s = (unsigned __int128)a->l[i] - b->l[i] - br;
    d[i] = (uint64_t)s;
    br = (s >> 64) & 1;
  }
  uint64_t mask = (uint64_t)0 - (uint64_t)br;
  unsigned __int128 c = 0;
  for (int i = 0; i < NL; i++) {
    c += (unsigned __int128)d[i] + (K(_P)[i] & mask);
    r->l[i] = (uint64_t)c;
    c >>= 64;
  }
}

static inline void FN(fp_neg)(FN(fp)* r, const FN(fp)* a) {
  FN(fp) z;
  FN(fp_zero)(&z);
  FN(fp_sub)(r, &z, a);
}

static inline void FN(fp_dbl)(FN(fp)* r, const FN(fp)* a) { FN(fp_add)(r, a, a); }

/* CIOS Montgomery multiplication, R = 2^(64 NL) */
static void FN(fp_mul)(FN(fp)* r, const FN(fp)* a, const FN(fp)* b) {
  uint64_t t[NL + 2];
  for (int i = 0; i < NL + 2; i++) t[i] = 0;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 c = 0;
    for (int j = 0; j < NL; j++) {
      unsigned __int128 acc = (unsigned __int128)a->l[j] * b->l[i] + t[j] + c;
      t[j] = (uint64_t)acc;
      c = acc >> 64;
    }
    unsigned __int128 acc = (unsigned __int128)t[NL] + c;
    t[NL] = (uint64_t)acc;
    t[NL + 1] = (uint64_t)(acc >> 64);
    uint64_t m = t[0] * K(_INV);
    acc = (unsigned __int128)m * K(_P)[0] + t[0];
    c = acc >> 64;
    for (int j = 1; j < NL; j++) {
      acc = (unsigned __int128)m * K(_P)[j] + t[j] + c;
      t[j - 1] = (uint64_t)acc;
      c = acc >> 64;
    }
    acc = (unsigned __int128)t[NL] + c;
    t[NL - 1] = (uint64_t)acc;
    t[NL] = t[NL + 1] + (uint64_t)(acc >> 64);
  }
  uint64_t d[NL];
  unsigned __int128 br = 0;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 s = (unsigned __int128)t[i] - K(_P)[i] - br;
    d[i] = (uint64_t)s;
    br = (s >> 64) & 1;
  }
  int ge = (t[NL] != 0) | (br == 0);
  for (int i = 0; i < NL; i++) r->l[i] = ge ? d[i] : t[i];
}

static inline void FN(fp_sqr)(FN(fp)* r, const FN(fp)* a) { FN(fp_mul)(r, a, a); }

static void FN(fp_mul_small)(FN(fp)* r, const FN(fp)* a, int k) {
  FN(fp) acc, cur = *a;
  FN(fp_zero)(&acc);
  while (k) {
    if (k & 1) FN(fp_add)(&acc, &acc, &cur);
    FN(fp_dbl)(&cur, &cur);
    k >>= 1;
  }
  *r = acc;
}

/* a^(p-2) */
static void FN(fp_inv)(FN(fp)* r, const FN(fp)* a) {
  uint64_t e[NL];
  unsigned __int128 br = 2;
  for (int i = 0; i < NL; i++) {
    unsigned __int128 s = (unsigned __int128)K(_P)[i] - br;
    e[i] = (uint64_t)s;
    br = (s >> 64) & 1;
  }
  FN(fp) acc;
  FN(fp_one)(&acc);
  for (int i = NL * 64 - 1; i >= 0; i--) {
    FN(fp_sqr)(&acc, &acc);
    if ((e[i >> 6] >> (i & 63)) & 1) FN(fp_mul)(&acc, &acc, a);
  }
  *r = acc;
}

static void FN(fp_halve)(FN(fp)* r, const FN(fp)* a) {
  uint64_t mask = (uint64_t)0 - (a->l[0] & 1), t[NL];
  unsigned __int128 c = 0;
  for (int i = 0; i < NL; i++) {
    c += (unsigned __int128)a->l[i] + (K(_P)[i] & mask);
    t[i] = (uint64_t)c;
    c >>= 64;
  }
  for (int i = 0; i < NL - 1; i++) r->l[i] = (t[i] >> 1) | (t[i + 1] << 63);
  r->l[NL - 1] = (t[NL - 1] >> 1) | ((uint64_t)c << 63);
}

static void FN(fp_mul_beta)(FN(fp)* r, const FN(fp)* a) {
  if (K(_BETA) == -1) {
    FN(fp_neg)(r, a);
  } else {
    FN(fp) t;
    FN(fp_mul_small)(&t, a, -K(_BETA));
    FN(fp_neg)(r, &t);
  }
}

/* ---------------------------------------------------------------- Fp2 */
static inline void FN(fp2_zero)(FN(fp2)* r) { memset(r, 0, sizeof(*r)); }
static inline void FN(fp2_one)(FN(fp2)* r) { FN(fp_one)(&r->c0); FN(fp_zero)(&r->c1); }
static inline int FN(fp2_is_zero)(const FN(fp2)* a) { return FN(fp_is_zero)(&a->c0) && FN(fp_is_zero)(&a->c1); }
static inline int FN(fp2_eq)(const FN(fp2)* a, const FN(fp2)* b) { return memcmp(a, b, sizeof(*a)) == 0; }
static inline void FN(fp2_add)(FN(fp2)* r, const FN(fp2)* a, const FN(fp2)* b) { FN(fp_add)(&r->c0, &a->c0, &b->c0); FN(fp_add)(&r->c1, &a->c1, &b->c1); }
static inline void FN(fp2_sub)(FN(fp2)* r, const FN(fp2)* a, const FN(fp2)* b) { FN(fp_sub)(&r->c0, &a->c0, &b->c0); FN(fp_sub)(&r->c1, &a->c1, &b->c1); }
static inline void FN(fp2_dbl)(FN(fp2)* r, const FN(fp2)* a) { FN(fp_dbl)(&r->c0, &a->c0); FN(fp_dbl)(&r->c1, &a->c1); }
static inline void FN(fp2_neg)(FN(fp2)* r, const FN(fp2)* a) { FN(fp_neg)(&r->c0, &a->c0); FN(fp_neg)(&r->c1, &a->c1); }
static inline void FN(fp2_conj)(FN(fp2)* r, const FN(fp2)* a) { r->c0 = a->c0; FN(fp_neg)(&r->c1, &a->c1); }
static inline void FN(fp2_halve)(FN(fp2)* r, const FN(fp2)* a) { FN(fp_halve)(&r->c0, &a->c0); FN(fp_halve)(&r->c1, &a->c1); }

static void FN(fp2_mul)(FN(fp2)* r, const FN(fp2)* a, const FN(fp2)* b) {
  FN(fp) t0, t1, t2, s0, s1;
  FN(fp_mul)(&t0, &a->c0, &b->c0);
  FN(fp_mul)(&t1, &a->c1, &b->c1);
  FN(fp_add)(&s0, &a->c0, &a->c1);
  FN(fp_add)(&s1, &b->c0, &b->c1);
  FN(fp_mul)(&t2, &s0, &s1);
  FN(fp_sub)(&t2, &t2, &t0);
  FN(fp_sub)(&r->c1, &t2, &t1);
  FN(fp_mul_beta)(&t1, &t1);
  FN(fp_add)(&r->c0, &t0, &t1);
}
static void FN(fp2_sqr)(FN(fp2)* r, const FN(fp2)* a) { FN(fp2) t = *a; FN(fp2_mul)(r, &t, &t); }
static void FN(fp2_mul_fp)(FN(fp2)* r, const FN(fp2)* a, const FN(fp)* k) { FN(fp_mul)(&r->c0, &a->c0, k); FN(fp_mul)(&r->c1, &a->c1, k); }
static void FN(fp2_mul_xi)(FN(fp2)* r, const FN(fp2)* a) {
  FN(fp) t0, t1, n0, n1;
  FN(fp_mul_small)(&t0, &a->c0, K(_XI0));
  FN(fp_mul_small)(&t1, &a->c1, K(_XI1));
  FN(fp_mul_beta)(&t1, &t1);
  FN(fp_add)(&n0, &t0, &t1);
  FN(fp_mul_small)(&t0, &a->c0, K(_XI1));
  FN(fp_mul_small)(&t1, &a->c1, K(_XI0));
  FN(fp_add)(&n1, &t0, &t1);
  r->c0 = n0;
  r->c1 = n1;
}
static void FN(fp2_inv)(FN(fp2)* r, const FN(fp2)* a) {
  FN(fp) t0, t1, n;
  FN(fp_sqr)(&t0, &a->c0);
  FN(fp_sqr)(&t1, &a->c1);
  FN(fp_mul_beta)(&t1, &t1);
  FN(fp_sub)(&n, &t0, &t1);
  FN(fp_inv)(&n, &n);
  FN(fp_mul)(&r->c0, &a->c0, &n);
  FN(fp_mul)(&t0, &a->c1, &n);
  FN(fp_neg)(&r->c1, &t0);
}

/* ---------------------------------------------------------------- Fp6 / Fp12 */
static void FN(fp6_add)(FN(fp6)* r, const FN(fp6)* a, const FN(fp6)* b) { FN(fp2_add)(&r->c0, &a->c0, &b->c0); FN(fp2_add)(&r->c1, &a->c1, &b->c1); FN(fp2_add)(&r->c2, &a->c2, &b->c2); }
static void FN(fp6_sub)(FN(fp6)* r, const FN(fp6)* a, const FN(fp6)* b) { FN(fp2_sub)(&r->c0, &a->c0, &b->c0); FN(fp2_sub)(&r->c1, &a->c1, &b->c1); FN(fp2_sub)(&r->c2, &a->c2, &b->c2); }
static void FN(fp6_neg)(FN(fp6)* r, const FN(fp6)* a) { FN(fp2_neg)(&r->c0, &a->c0); FN(fp2_neg)(&r->c1, &a->c1); FN(fp2_neg)(&r->c2, &a->c2); }
static void FN(fp6_mul_v)(FN(fp6)* r, const FN(fp6)* a) {
  FN(fp2) t;
  FN(fp2_mul_xi)(&t, &a->c2);
  r->c2 = a->c1;
  r->c1 = a->c0;
  r->c0 = t;
}
/* schoolbook (9 Fp2 multiplications): deliberately not the Karatsuba form the kernels use */
static void FN(fp6_mul)(FN(fp6)* r, const FN(fp6)* a, const FN(fp6)* b) {
  FN(fp2) a0b0, a0b1, a0b2, a1b0, a1b1, a1b2, a2b0, a2b1, a2b2, t, x0, x1, x2;
  FN(fp2_mul)(&a0b0, &a->c0, &b->c0); FN(fp2_mul)(&a0b1, &a->c0, &b->c1); FN(fp2_mul)(&a0b2, &a->c0, &b->c2);
  FN(fp2_mul)(&a1b0, &a->c1, &b->c0); FN(fp2_mul)(&a1b1, &a->c1, &b->c1); FN(fp2_mul)(&a1b2, &a->c1, &b->c2);
  FN(fp2_mul)(&a2b0, &a->c2, &b->c0); FN(fp2_mul)(&a2b1, &a->c2, &b->c1); FN(fp2_mul)(&a2b2, &a->c2, &b->c2);
  FN(fp2_add)(&t, &a1b2, &a2b1); FN(fp2_mul_xi)(&t, &t); FN(fp2_add)(&x0, &a0b0, &t);
  FN(fp2_mul_xi)(&t, &a2b2); FN(fp2_add)(&x1, &a0b1, &a1b0); FN(fp2_add)(&x1, &x1, &t);
  FN(fp2_add)(&x2, &a0b2, &a1b1); FN(fp2_add)(&x2, &x2, &a2b0);
  r->c0 = x0; r->c1 = x1; r->c2 = x2;
}
static void FN(fp6_inv)(FN(fp6)* r, const FN(fp6)* a) {
  FN(fp2) c0, c1, c2, t, u;
  FN(fp2_sqr)(&c0, &a->c0); FN(fp2_mul)(&t, &a->c1, &a->c2); FN(fp2_mul_xi)(&t, &t); FN(fp2_sub)(&c0, &c0, &t);
  FN(fp2_sqr)(&c1, &a->c2); FN(fp2_mul_xi)(&c1, &c1); FN(fp2_mul)(&t, &a->c0, &a->c1); FN(fp2_sub)(&c1, &c1, &t);
  FN(fp2_sqr)(&c2, &a->c1); FN(fp2_mul)(&t, &a->c0, &a->c2); FN(fp2_sub)(&c2, &c2, &t);
  FN(fp2_mul)(&t, &a->c2, &c1); FN(fp2_mul)(&u, &a->c1, &c2); FN(fp2_add)(&t, &t, &u); FN(fp2_mul_xi)(&t, &t);
  FN(fp2_mul)(&u, &a->c0, &c0); FN(fp2_add)(&t, &t, &u);
  FN(fp2_inv)(&t, &t);
  FN(fp2_mul)(&r->c0, &c0, &t); FN(fp2_mul)(&r->c1, &c1, &t); FN(fp2_mul)(&r->c2, &c2, &t);
}

static void FN(fp12_one)(FN(fp12)* r) { memset(r, 0, sizeof(*r)); FN(fp_one)(&r->c0.c0.c0); }
static void FN(fp12_conj)(FN(fp12)* r, const FN(fp12)* a) { r->c0 = a->c0; FN(fp6_neg)(&r->c1, &a->c1); }
static void FN(fp12_mul)(FN(fp12)* r, const FN(fp12)* a, const FN(fp12)* b) {
  FN(fp6) t0, t1, s0, s1, x;
  FN(fp6_mul)(&t0, &a->c0, &b->c0);
  FN(fp6_mul)(&t1, &a->c1, &b->c1);
  FN(fp6_add)(&s0, &a->c0, &a->c1);
  FN(fp6_add)(&s1, &b->c0, &b->c1);
  FN(fp6_mul)(&x, &s0, &s1);
  FN(fp6_sub)(&x, &x, &t0);
  FN(fp6_sub)(&r->c1, &x, &t1);
  FN(fp6_mul_v)(&t1, &t1);
  FN(fp6_add)(&r->c0, &t0, &t1);
}
static void FN(fp12_sqr)(FN(fp12)* r, const FN(fp12)* a) { FN(fp12) t = *a; FN(fp12_mul)(r, &t, &t); }
static void FN(fp12_inv)(FN(fp12)* r, const FN(fp12)* a) {
  FN(fp6) t0, t1;
  FN(fp6_mul)(&t0, &a->c0, &a->c0);
  FN(fp6_mul)(&t1, &a->c1, &a->c1);
  FN(fp6_mul_v)(&t1, &t1);
  FN(fp6_sub)(&t0, &t0, &t1);
  FN(fp6_inv)(&t0, &t0);
  FN(fp6_mul)(&r->c0, &a->c0, &t0);
  FN(fp6_mul)(&t1, &a->c1, &t0);
  FN(fp6_neg)(&r->c1, &t1);
}
static void FN(fp12_frob)(FN(fp12)* r, const FN(fp12)* a, int k) {
  const FN(fp2)* src[6] = {&a->c0.c0, &a->c1.c0, &a->c0.c1, &a->c1.c1, &a->c0.c2, &a->c1.c2};
  FN(fp2)* dst[6] = {&r->c0.c0, &r->c1.c0, &r->c0.c1, &r->c1.c1, &r->c0.c2, &r->c1.c2};
  FN(fp12) out;
  FN(fp2)* od[6] = {&out.c0.c0, &out.c1.c0, &out.c0.c1, &out.c1.c1, &out.c0.c2, &out.c1.c2};
  for (int i = 0; i < 6; i++) {
    FN(fp2) x, g;
    if (k & 1) FN(fp2_conj)(&x, src[i]); else x = *src[i];
    const uint64_t (*G)[2][NL] = k == 1 ? K(_GAMMA1) : (k == 2 ? K(_GAMMA2) : K(_GAMMA3));
    memcpy(g.c0.l, G[i][0], sizeof(g.c0.l));
    memcpy(g.c1.l, G[i][1], sizeof(g.c1.l));
    FN(fp2_mul)(od[i], &x, &g);
  }
  for (int i = 0; i < 6; i++) *dst[i] = *od[i];
}

/* ---------------------------------------------------------------- pairing */
typedef struct { FN(fp) x, y; } FN(g1a);
typedef struct { FN(fp2) x, y; } FN(g2a);
typedef struct { FN(fp2) x, y, z; } FN(g2p);

static int FN(g1a_is_inf)(const FN(g1a)* p) { return FN(fp_is_zero)(&p->x) && FN(fp_is_zero)(&p->y); }
static int FN(g2a_is_inf)(const FN(g2a)* p) { return FN(fp2_is_zero)(&p->x) && FN(fp2_is_zero)(&p->y); }

/* multiply f by the sparse line  (M-twist: c0 + c1 v + c4 v w ; D-twist: c0 + c3 w + c4 v w) via a full Fp12 mul */
static void FN(mul_line)(FN(fp12)* f, const FN(fp2)* r0, const FN(fp2)* r1, const FN(fp2)* r2, const FN(fp)* px, const FN(fp)* py) {
  FN(fp12) l;
  FN(fp2) a, b;
  memset(&l, 0, sizeof(l));
  FN(fp2_mul_fp)(&a, r0, py);
  FN(fp2_mul_fp)(&b, r1, px);
  if (K(_MTWIST)) { l.c0.c0 = *r2; l.c0.c1 = b; l.c1.c1 = a; }
  else { l.c0.c0 = a; l.c1.c0 = b; l.c1.c1 = *r2; }
  FN(fp12_mul)(f, f, &l);
}

static void FN(dbl_step)(FN(g2p)* T, FN(fp2)* r0, FN(fp2)* r1, FN(fp2)* r2) {
  FN(fp2) A, B, C, E, F, G, H, I, J, EE, t, b3;
  memcpy(b3.c0.l, K(_B3TW)[0], sizeof(b3.c0.l));
  memcpy(b3.c1.l, K(_B3TW)[1], sizeof(b3.c1.l));
  FN(fp2_mul)(&A, &T->x, &T->y); FN(fp2_halve)(&A, &A);
  FN(fp2_sqr)(&B, &T->y);
  FN(fp2_sqr)(&C, &T->z);
  FN(fp2_mul)(&E, &C, &b3);
  FN(fp2_dbl)(&F, &E); FN(fp2_add)(&F, &F, &E);
  FN(fp2_add)(&G, &B, &F); FN(fp2_halve)(&G, &G);
  FN(fp2_add)(&H, &T->y, &T->z); FN(fp2_sqr)(&H, &H); FN(fp2_add)(&t, &B, &C); FN(fp2_sub)(&H, &H, &t);
  FN(fp2_sub)(&I, &E, &B);
  FN(fp2_sqr)(&J, &T->x);
  FN(fp2_sqr)(&EE, &E);
  FN(fp2_sub)(&t, &B, &F); FN(fp2_mul)(&T->x, &A, &t);
  FN(fp2_sqr)(&G, &G); FN(fp2_dbl)(&t, &EE); FN(fp2_add)(&t, &t, &EE); FN(fp2_sub)(&T->y, &G, &t);
  FN(fp2_mul)(&T->z, &B, &H);
  FN(fp2_neg)(r0, &H);
  FN(fp2_dbl)(r1, &J); FN(fp2_add)(r1, r1, &J);
  *r2 = I;
}

static void FN(add_step)(FN(g2p)* T, const FN(fp2)* qx, const FN(fp2)* qy, FN(fp2)* r0, FN(fp2)* r1, FN(fp2)* r2) {
  FN(fp2) O, L, C, D, E, F, G, H, t, t2;
  FN(fp2_mul)(&t, qy, &T->z); FN(fp2_sub)(&O, &T->y, &t);
  FN(fp2_mul)(&t, qx, &T->z); FN(fp2_sub)(&L, &T->x, &t);
  FN(fp2_sqr)(&C, &O); FN(fp2_sqr)(&D, &L); FN(fp2_mul)(&E, &L, &D); FN(fp2_mul)(&F, &T->z, &C); FN(fp2_mul)(&G, &T->x, &D);
  FN(fp2_dbl)(&t, &G); FN(fp2_add)(&H, &E, &F); FN(fp2_sub)(&H, &H, &t);
  FN(fp2_mul)(&t2, &T->y, &E);
  FN(fp2_mul)(&T->x, &L, &H);
  FN(fp2_sub)(&t, &G, &H); FN(fp2_mul)(&t, &O, &t); FN(fp2_sub)(&T->y, &t, &t2);
  FN(fp2_mul)(&T->z, &T->z, &E);
  FN(fp2_mul)(&t, &L, qy); FN(fp2_mul)(&t2, qx, &O); FN(fp2_sub)(r2, &t2, &t);
  *r0 = L;
  FN(fp2_neg)(r1, &O);
}

static void FN(miller_loop)(FN(fp12)* f, const FN(g1a)* P, const FN(g2a)* Q, int n_pairs) {
  FN(fp12_one)(f);
  FN(g2p) T[8];
  int live[8], any = 0;
  if (n_pairs > 8) n_pairs = 8;
  for (int k = 0; k < n_pairs; k++) {
    live[k] = !(FN(g1a_is_inf)(&P[k]) || FN(g2a_is_inf)(&Q[k]));
    T[k].x = Q[k].x; T[k].y = Q[k].y; FN(fp2_one)(&T[k].z);
    any |= live[k];
  }
  if (!any) return;
  FN(fp2) r0, r1, r2;
  for (int i = K(_ATE_BITS) - 2; i >= 0; i--) {
    FN(fp12_sqr)(f, f);
    int bit = i >= 64 ? (int)((K(_ATE_HI) >> (i - 64)) & 1) : (int)((K(_ATE_LO) >> i) & 1);
    for (int k = 0; k < n_pairs; k++) {
      if (!live[k]) continue;
      FN(dbl_step)(&T[k], &r0, &r1, &r2);
      FN(mul_line)(f, &r0, &r1, &r2, &P[k].x, &P[k].y);
      if (bit) {
        FN(add_step)(&T[k], &Q[k].x, &Q[k].y, &r0, &r1, &r2);
        FN(mul_line)(f, &r0, &r1, &r2, &P[k].x, &P[k].y);
      }
    }
  }
  if (K(_IS_BN)) {
    for (int k = 0; k < n_pairs; k++) {
      if (!live[k]) continue;
      FN(fp2) x1, y1, x2, y2, g;
      FN(fp2_conj)(&x1, &Q[k].x); memcpy(g.c0.l, K(_GAMMA1)[2][0], sizeof(g.c0.l)); memcpy(g.c1.l, K(_GAMMA1)[2][1], sizeof(g.c1.l)); FN(fp2_mul)(&x1, &x1, &g);
      FN(fp2_conj)(&y1, &Q[k].y); memcpy(g.c0.l, K(_GAMMA1)[3][0], sizeof(g.c0.l)); memcpy(g.c1.l, K(_GAMMA1)[3][1], sizeof(g.c1.l)); FN(fp2_mul)(&y1, &y1, &g);
      memcpy(g.c0.l, K(_GAMMA2)[2][0], sizeof(g.c0.l)); memcpy(g.c1.l, K(_GAMMA2)[2][1], sizeof(g.c1.l)); FN(fp2_mul)(&x2, &Q[k].x, &g);
      memcpy(g.c0.l, K(_GAMMA2)[3][0], sizeof(g.c0.l)); memcpy(g.c1.l, K(_GAMMA2)[3][1], sizeof(g.c1.l)); FN(fp2_mul)(&y2, &Q[k].y, &g);
      FN(fp2_neg)(&y2, &y2);
      FN(add_step)(&T[k], &x1, &y1, &r0, &r1, &r2);
      FN(mul_line)(f, &r0, &r1, &r2, &P[k].x, &P[k].y);
      FN(add_step)(&T[k], &x2, &y2, &r0, &r1, &r2);
      FN(mul_line)(f, &r0, &r1, &r2, &P[k].x, &P[k].y);
    }
  }
  if (K(_X_NEG)) FN(fp12_conj)(f, f);
}

/* z^|x| (plain squarings: the oracle does not use cyclotomic squaring), conjugated for a negative seed */
static void FN(fp12_expt)(FN(fp12)* r, const FN(fp12)* z) {
  FN(fp12) acc = *z;
  int top = 63;
  while (!((K(_X_ABS) >> top) & 1)) top--;
  for (int i = top - 1; i >= 0; i--) {
    FN(fp12_sqr)(&acc, &acc);
    if ((K(_X_ABS) >> i) & 1) FN(fp12_mul)(&acc, &acc, z);
  }
  if (K(_X_NEG)) FN(fp12_conj)(&acc, &acc);
  *r = acc;
}

static void FN(final_exp)(FN(fp12)* out, const FN(fp12)* f) {
  FN(fp12) r, t0, t1, t2;
  FN(fp12_conj)(&t0, f);
  FN(fp12_inv)(&t1, f);
  FN(fp12_mul)(&t0, &t0, &t1);
  FN(fp12_frob)(&t1, &t0, 2);
  FN(fp12_mul)(&r, &t1, &t0);
  if (!K(_IS_BN)) {
    FN(fp12_sqr)(&t0, &r);
    FN(fp12_expt)(&t1, &r);
    FN(fp12_conj)(&t2, &r);
    FN(fp12_mul)(&t1, &t1, &t2);
    FN(fp12_expt)(&t2, &t1);
    FN(fp12_conj)(&t1, &t1);
    FN(fp12_mul)(&t1, &t1, &t2);
    FN(fp12_expt)(&t2, &t1);
    FN(fp12_frob)(&t1, &t1, 1);
    FN(fp12_mul)(&t1, &t1, &t2);
    FN(fp12_mul)(&r, &r, &t0);
    FN(fp12_expt)(&t0, &t1);
    FN(fp12_expt)(&t2, &t0);
    FN(fp12_frob)(&t0, &t1, 2);
    FN(fp12_conj)(&t1, &t1);
    FN(fp12_mul)(&t1, &t1, &t2);
    FN(fp12_mul)(&t1, &t1, &t0);
    FN(fp12_mul)(out, &r, &t1);
  } else {
    FN(fp12) fx, f2x, f6x, f6x2, f12x3, a, b;
    FN(fp12_expt)(&fx, &r);
    FN(fp12_sqr)(&f2x, &fx);
    FN(fp12_sqr)(&t0, &f2x);
    FN(fp12_mul)(&f6x, &t0, &f2x);
    FN(fp12_expt)(&f6x2, &f6x);
    FN(fp12_sqr)(&t0, &f6x2);
    FN(fp12_expt)(&f12x3, &t0);
    FN(fp12_mul)(&a, &f12x3, &f6x2);
    FN(fp12_mul)(&a, &a, &f6x);
    FN(fp12_conj)(&t0, &f2x);
    FN(fp12_mul)(&b, &a, &t0);
    FN(fp12_mul)(&t0, &a, &f6x2);
    FN(fp12_mul)(&t0, &t0, &r);
    FN(fp12_frob)(&t1, &b, 1);
    FN(fp12_mul)(&t0, &t0, &t1);
    FN(fp12_frob)(&t1, &a, 2);
    FN(fp12_mul)(&t0, &t0, &t1);
    FN(fp12_conj)(&t1, &r);
    FN(fp12_mul)(&t1, &b, &t1);
    FN(fp12_frob)(&t2, &t1, 3);
    FN(fp12_mul)(out, &t0, &t2);
  }
}
