/* oracle/cref_ec.h -- TEST INFRASTRUCTURE (oracle): Jacobian group arithmetic, Pippenger MSM and input
 * generation over one coordinate field.  Included twice per curve by cref.c:
 *   EF(name) = name##_g1<suf> with the coordinate field Fp, and name##_g2<suf> with Fp2.
 * Macros expected: EF(), ET (coordinate type), E_add/E_sub/E_mul/E_sqr/E_dbl/E_neg/E_inv/E_is_zero/E_eq/E_one/E_zero.
 *
 * Semantics restated: MultiScalarMul = sum_i [s_i]P_i with affine infinity = (0,0) and a Jacobian
 * accumulator, then one conversion to affine (driver/gurvy/bls12381/bls12-381.go:766-783; the naive
 * definition in driver/kilic/bls12-381.go:247-254).  Jacobian (not XYZZ) on purpose: an independent
 * formula set from the kernels'.
 */
typedef struct { ET x, y; } EF(aff);
typedef struct { ET x, y, z; } EF(jac);

static int EF(aff_is_inf)(const EF(aff)* p) { return E_is_zero(&p->x) && E_is_zero(&p->y); }
static void EF(jac_set_inf)(EF(jac)* r) { E_one(&r->x); E_one(&r->y); E_zero(&r->z); }
static int EF(jac_is_inf)(const EF(jac)* p) { return E_is_zero(&p->z); }

static void EF(jac_dbl)(EF(jac)* r, const EF(jac)* p) {
  if (EF(jac_is_inf)(p)) { EF(jac_set_inf)(r); return; }
  ET A, B, C, D, E, F, t, X3, Y3, Z3;
  E_sqr(&A, &p->x);
  E_sqr(&B, &p->y);
  E_sqr(&C, &B);
  E_add(&t, &p->x, &B); E_sqr(&t, &t); E_sub(&t, &t, &A); E_sub(&t, &t, &C); E_dbl(&D, &t);
  E_dbl(&E, &A); E_add(&E, &E, &A);
  E_sqr(&F, &E);
  E_dbl(&t, &D); E_sub(&X3, &F, &t);
  E_sub(&t, &D, &X3); E_mul(&Y3, &E, &t);
  E_dbl(&t, &C); E_dbl(&t, &t); E_dbl(&t, &t); E_sub(&Y3, &Y3, &t);
  E_mul(&Z3, &p->y, &p->z); E_dbl(&Z3, &Z3);
  r->x = X3; r->y = Y3; r->z = Z3;
}

/* r = p + q (q affine, optionally negated) */
static void EF(jac_madd)(EF(jac)* r, const EF(jac)* p, const EF(aff)* q_in, int negate) {
  if (EF(aff_is_inf)(q_in)) { *r = *p; return; }
  EF(aff) q = *q_in;
  if (negate) E_neg(&q.y, &q.y);
  if (EF(jac_is_inf)(p)) { r->x = q.x; r->y = q.y; E_one(&r->z); return; }
  ET Z1Z1, U2, S2, H, HH, HHH, Rr, V, t, X3, Y3, Z3;
  E_sqr(&Z1Z1, &p->z);
  E_mul(&U2, &q.x, &Z1Z1);
  E_mul(&S2, &q.y, &p->z); E_mul(&S2, &S2, &Z1Z1);
  if (E_eq(&U2, &p->x)) {
    if (E_eq(&S2, &p->y)) { EF(jac_dbl)(r, p); return; }
    EF(jac_set_inf)(r);
    return;
  }
  E_sub(&H, &U2, &p->x);
  E_sub(&Rr, &S2, &p->y);
  E_sqr(&HH, &H);
  E_mul(&HHH, &H, &HH);
  E_mul(&V, &p->x, &HH);
  E_sqr(&X3, &Rr); E_sub(&X3, &X3, &HHH); E_sub(&X3, &X3, &V); E_sub(&X3, &X3, &V);
  E_sub(&t, &V, &X3); E_mul(&Y3, &Rr, &t); E_mul(&t, &p->y, &HHH); E_sub(&Y3, &Y3, &t);
  E_mul(&Z3, &p->z, &H);
  r->x = X3; r->y = Y3; r->z = Z3;
}

/* r = p + q (both Jacobian) */
static void EF(jac_add)(EF(jac)* r, const EF(jac)* p, const EF(jac)* q) {
  if (EF(jac_is_inf)(q)) { *r = *p; return; }
  if (EF(jac_is_inf)(p)) { *r = *q; return; }
  ET Z1Z1, Z2Z2, U1, U2, S1, S2, H, HH, HHH, Rr, V, t, X3, Y3, Z3;
  E_sqr(&Z1Z1, &p->z); E_sqr(&Z2Z2, &q->z);
  E_mul(&U1, &p->x, &Z2Z2); E_mul(&U2, &q->x, &Z1Z1);
  E_mul(&S1, &p->y, &q->z); E_mul(&S1, &S1, &Z2Z2);
  E_mul(&S2, &q->y, &p->z); E_mul(&S2, &S2, &Z1Z1);
  if (E_eq(&U1, &U2)) {
    if (E_eq(&S1, &S2)) { EF(jac_dbl)(r, p); return; }
    EF(jac_set_inf)(r);
    return;
  }
  E_sub(&H, &U2, &U1); E_sub(&Rr, &S2, &S1);
  E_sqr(&HH, &H); E_mul(&HHH, &H, &HH); E_mul(&V, &U1, &HH);
  E_sqr(&X3, &Rr); E_sub(&X3, &X3, &HHH); E_sub(&X3, &X3, &V); E_sub(&X3, &X3, &V);
  E_sub(&t, &V, &X3); E_mul(&Y3, &Rr, &t); E_mul(&t, &S1, &HHH); E_sub(&Y3, &Y3, &t);
  E_mul(&Z3, &p->z, &q->z); E_mul(&Z3, &Z3, &H);
  r->x = X3; r->y = Y3; r->z = Z3;
}

static void EF(jac_to_aff)(EF(aff)* r, const EF(jac)* p) {
  if (EF(jac_is_inf)(p)) { E_zero(&r->x); E_zero(&r->y); return; }
  ET zi, zi2, zi3;
  E_inv(&zi, &p->z);
  E_sqr(&zi2, &zi);
  E_mul(&zi3, &zi2, &zi);
  E_mul(&r->x, &p->x, &zi2);
  E_mul(&r->y, &p->y, &zi3);
}

/* n Jacobian points -> affine with one inversion (Montgomery's trick); infinities stay (0,0) */
static void EF(batch_to_aff)(EF(aff)* out, const EF(jac)* in, size_t n) {
  ET* pre = (ET*)malloc((n ? n : 1) * sizeof(ET));
  ET acc, inv;
  E_one(&acc);
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    if (!EF(jac_is_inf)(&in[i])) E_mul(&acc, &acc, &in[i].z);
  }
  E_inv(&inv, &acc);
  for (size_t i = n; i-- > 0;) {
    if (EF(jac_is_inf)(&in[i])) { E_zero(&out[i].x); E_zero(&out[i].y); continue; }
    ET zi, zi2, zi3;
    E_mul(&zi, &inv, &pre[i]);
    E_mul(&inv, &inv, &in[i].z);
    E_sqr(&zi2, &zi);
    E_mul(&zi3, &zi2, &zi);
    E_mul(&out[i].x, &in[i].x, &zi2);
    E_mul(&out[i].y, &in[i].y, &zi3);
  }
  free(pre);
}

/* [k]P, k = 4 little-endian 64-bit words (already canonical) */
static void EF(scalar_mul)(EF(jac)* r, const EF(aff)* p, const uint64_t k[4]) {
  EF(jac) acc;
  EF(jac_set_inf)(&acc);
  for (int i = 255; i >= 0; i--) {
    EF(jac_dbl)(&acc, &acc);
    if ((k[i >> 6] >> (i & 63)) & 1) EF(jac_madd)(&acc, &acc, p, 0);
  }
  *r = acc;
}

/* ---- Pippenger with signed digits; one task = (window, point range) ------------------------------------ */
typedef struct {
  const EF(aff)* points;
  const uint64_t* scalars; /* canonical, 4 words each */
  size_t n;
  int c, W;
  /* task queue */
  int n_tasks, n_splits;
  volatile int next_task;
  EF(jac)* partial; /* [W][n_splits] */
} EF(msm_job);

static void EF(msm_task)(EF(msm_job)* job, int task) {
  const int w = task / job->n_splits, sp = task % job->n_splits;
  const size_t lo = job->n * (size_t)sp / job->n_splits, hi = job->n * (size_t)(sp + 1) / job->n_splits;
  const int c = job->c;
  const uint32_t half = 1u << (c - 1);
  EF(jac)* buckets = (EF(jac)*)malloc((size_t)half * sizeof(EF(jac)));
  for (uint32_t b = 0; b < half; b++) EF(jac_set_inf)(&buckets[b]);
  for (size_t i = lo; i < hi; i++) {
    /* digit w of scalar i: recompute the carry chain up to window w */
    const uint64_t* s = job->scalars + 4 * i;
    uint32_t carry = 0, v = 0;
    int neg = 0;
    for (int ww = 0; ww <= w; ww++) {
      int bit = ww * c;
      v = 0;
      if (bit < 256) {
        int word = bit >> 6, sh = bit & 63;
        unsigned __int128 two = s[word];
        if (word + 1 < 4) two |= (unsigned __int128)s[word + 1] << 64;
        v = (uint32_t)((two >> sh) & ((1u << c) - 1));
      }
      v += carry;
      if (v > half) { v = (1u << c) - v; neg = 1; carry = 1; } else { neg = 0; carry = 0; }
    }
    if (v == 0) continue;
    EF(jac_madd)(&buckets[v - 1], &buckets[v - 1], &job->points[i], neg);
  }
  EF(jac) run, sum;
  EF(jac_set_inf)(&run);
  EF(jac_set_inf)(&sum);
  for (uint32_t b = half; b-- > 0;) {
    EF(jac_add)(&run, &run, &buckets[b]);
    EF(jac_add)(&sum, &sum, &run);
  }
  job->partial[task] = sum;
  free(buckets);
}

static void* EF(msm_worker)(void* arg) {
  EF(msm_job)* job = (EF(msm_job)*)arg;
  for (;;) {
    int t = __sync_fetch_and_add(&job->next_task, 1);
    if (t >= job->n_tasks) break;
    EF(msm_task)(job, t);
  }
  return NULL;
}

static int EF(msm)(const EF(aff)* points, const uint64_t* canon_scalars, size_t n, int c, int threads, EF(aff)* out) {
  if (n == 0) { E_zero(&out->x); E_zero(&out->y); return 0; }
  if (c <= 0) {
    c = 1;
    while (((size_t)1 << (c + 3)) < n && c < 16) c++;
    if (c < 2) c = 2;
  }
  EF(msm_job) job;
  job.points = points; job.scalars = canon_scalars; job.n = n; job.c = c;
  job.W = (CR_FR_BITS + 1 + c - 1) / c;
  if (threads < 1) threads = 1;
  job.n_splits = (threads + job.W - 1) / job.W;
  if ((size_t)job.n_splits > n) job.n_splits = (int)n;
  if (job.n_splits < 1) job.n_splits = 1;
  job.n_tasks = job.W * job.n_splits;
  job.next_task = 0;
  job.partial = (EF(jac)*)malloc((size_t)job.n_tasks * sizeof(EF(jac)));
  if (threads == 1) {
    EF(msm_worker)(&job);
  } else {
    pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, EF(msm_worker), &job);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th);
  }
  EF(jac) acc;
  EF(jac_set_inf)(&acc);
  for (int w = job.W - 1; w >= 0; w--) {
    for (int k = 0; k < c; k++) EF(jac_dbl)(&acc, &acc);
    for (int sp = 0; sp < job.n_splits; sp++) EF(jac_add)(&acc, &acc, &job.partial[w * job.n_splits + sp]);
  }
  EF(jac_to_aff)(out, &acc);
  free(job.partial);
  return 0;
}

/* synthetic inputs: P_i = [k0]G + i [k1]G, i = 0..n-1 (distinct subgroup points), batch-normalised */
static void EF(gen_points)(const EF(aff)* gen, const uint64_t k0[4], const uint64_t k1[4], size_t n, EF(aff)* out) {
  EF(jac) base, step;
  EF(aff) step_a;
  EF(scalar_mul)(&base, gen, k0);
  EF(scalar_mul)(&step, gen, k1);
  EF(jac_to_aff)(&step_a, &step);
  const size_t CH = 4096;
  EF(jac)* buf = (EF(jac)*)malloc(CH * sizeof(EF(jac)));
  for (size_t off = 0; off < n; off += CH) {
    size_t m = n - off < CH ? n - off : CH;
    for (size_t i = 0; i < m; i++) {
      buf[i] = base;
      EF(jac_madd)(&base, &base, &step_a, 0);
    }
    EF(batch_to_aff)(out + off, buf, m);
  }
  free(buf);
}
