/* Oracle B -- plain-C CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Restates, for the two Pasta curves, the algorithms of alex-ozdemir/bulletproofs that the
 * MI355X library replaces, following the reference line by line where it matters:
 *   naive msm            src/util/msm.rs:13-24
 *   pippenger_msm        src/util/msm.rs:88-193   (same window schedule, unsigned digits,
 *                        zero filter, scalar==1 shortcut, full Jacobian bucket adds,
 *                        running-sum reduction, high->low Horner; threads over windows
 *                        stand in for rayon at :119)
 *   ip                   src/util/mod.rs:17-23
 *   scalar / generator fold, Bp::prove   src/proofs/bp_iter.rs:49-128
 *   msm_bench input recipe               examples/msm_bench.rs:40-44
 *   scale_vec/sum_vecs k-ary folds, powers, hadamard, hadamard_gp, padding helpers   src/util/mod.rs:25-134
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (bulletproofs_amd + libbpgpu.so) never does.
 *
 * PARITY UNPINNED: the reference holds no known-answer vectors for this path and its Rust
 * toolchain / arkworks dependencies are absent here.  The arithmetic of
 * ark-ff / ark-ec 0.3.0, rand 0.8.4 / rand_chacha 0.3.1 and ark-marlin 0.3.0's
 * FiatShamirRng is restated from their published algorithms.  What pins this file: the
 * independent big-integer model oracle/pasta_model.py (tests/test_oracle.py), RFC 7539 /
 * RFC 7693 known answers for ChaCha20 / BLAKE2s, and the reference's own algebraic
 * invariants (IpaRelation::check, verifier acceptance, Pippenger == naive).
 *
 * Data layout at this API = ark-ff memory layout: Fp256 as 4 x u64 LE in Montgomery form
 * (R = 2^256); Jacobian point = X|Y|Z (96 B), identity has Z = 0.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;
typedef struct { fe x, y, z; } jac;
typedef struct {
  uint64_t p[4];
  uint64_t inv;   /* -p^-1 mod 2^64 */
  fe one;         /* R mod p */
  fe r2;          /* R^2 mod p */
  fe b;           /* curve constant 5 in Montgomery form (used when this is a base field) */
  /* Tonelli-Shanks constants: p - 1 = 2^32 * t */
  uint64_t t[4], t_plus1_half[4];
  fe root_of_unity; /* g^t for a non-residue g */
} field;

static field F[2];      /* 0: Pallas base (= Vesta scalar), 1: Pallas scalar (= Vesta base) */
static int g_init = 0;
static int g_threads = 1;

static const uint64_t P0[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0x0ull, 0x4000000000000000ull};
static const uint64_t P1[4] = {0x8c46eb2100000001ull, 0x224698fc0994a8ddull, 0x0ull, 0x4000000000000000ull};

/* ---- field --------------------------------------------------------------------- */
static int ge4(const uint64_t* a, const uint64_t* b) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
static void sub4(uint64_t* a, const uint64_t* b) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
static int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int fe_eq(const fe* a, const fe* b) { return memcmp(a, b, 32) == 0; }
static void fe_add(const field* f, fe* r, const fe* a, const fe* b) {
  u128 c = 0;
  fe t;
  for (int i = 0; i < 4; i++) {
    c += (u128)a->v[i] + b->v[i];
    t.v[i] = (uint64_t)c;
    c >>= 64;
  }
  if (ge4(t.v, f->p)) sub4(t.v, f->p);
  *r = t;
}
static void fe_sub(const field* f, fe* r, const fe* a, const fe* b) {
  fe t = *a;
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)t.v[i] - b->v[i] - br;
    t.v[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)t.v[i] + f->p[i];
      t.v[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  *r = t;
}
static void fe_neg(const field* f, fe* r, const fe* a) {
  fe z = {{0, 0, 0, 0}};
  if (fe_is_zero(a)) *r = z; else fe_sub(f, r, &z, a);
}
static void fe_dbl(const field* f, fe* r, const fe* a) { fe_add(f, r, a, a); }
static void fe_mul(const field* f, fe* r, const fe* a, const fe* b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) {
      c += (u128)a->v[j] * b->v[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * f->inv;
    c = (u128)m * f->p[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; j++) {
      c += (u128)m * f->p[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fe o = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || ge4(o.v, f->p)) sub4(o.v, f->p);
  *r = o;
}
static void fe_sqr(const field* f, fe* r, const fe* a) { fe_mul(f, r, a, a); }
static void fe_to_canon(const field* f, uint64_t out[4], const fe* a) {
  fe one = {{1, 0, 0, 0}}, r;
  fe_mul(f, &r, a, &one);
  memcpy(out, r.v, 32);
}
static void fe_from_canon(const field* f, fe* r, const uint64_t in[4]) {
  fe t;
  memcpy(t.v, in, 32);
  fe_mul(f, r, &t, &f->r2);
}
static void fe_pow(const field* f, fe* r, const fe* a, const uint64_t e[4]) {
  fe acc = f->one;
  int started = 0;
  for (int i = 3; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      if (started) fe_sqr(f, &acc, &acc);
      if ((e[i] >> b) & 1) {
        if (started) fe_mul(f, &acc, &acc, a); else { acc = *a; started = 1; }
      }
    }
  *r = acc;
}
static void fe_inv(const field* f, fe* r, const fe* a) {
  uint64_t e[4] = {f->p[0] - 2, f->p[1], f->p[2], f->p[3]};
  fe_pow(f, r, a, e);
}
/* Tonelli-Shanks; returns 0 if a is a non-residue */
static int fe_sqrt(const field* f, fe* r, const fe* a) {
  if (fe_is_zero(a)) { *r = *a; return 1; }
  fe w, x, b, z = f->root_of_unity;
  /* w = a^((t-1)/2); x = a*w = a^((t+1)/2); b = x*w = a^t */
  uint64_t e[4];
  memcpy(e, f->t, 32);           /* (t-1)/2 : t is odd */
  e[0] &= ~1ull;
  for (int i = 0; i < 3; i++) e[i] = (e[i] >> 1) | (e[i + 1] << 63);
  e[3] >>= 1;
  fe_pow(f, &w, a, e);
  if (e[0] == 0 && e[1] == 0 && e[2] == 0 && e[3] == 0) w = f->one;
  fe_mul(f, &x, a, &w);
  fe_mul(f, &b, &x, &w);
  int v = 32;
  while (!fe_eq(&b, &f->one)) {
    int k = 0;
    fe b2k = b;
    while (!fe_eq(&b2k, &f->one)) {
      fe_sqr(f, &b2k, &b2k);
      k++;
      if (k == v) return 0; /* non-residue */
    }
    fe ww = z;
    for (int j = 0; j < v - k - 1; j++) fe_sqr(f, &ww, &ww);
    fe_sqr(f, &z, &ww);
    fe_mul(f, &b, &b, &z);
    fe_mul(f, &x, &x, &ww);
    v = k;
  }
  *r = x;
  return 1;
}

static void field_init(field* f, const uint64_t p[4]) {
  memcpy(f->p, p, 32);
  uint64_t inv = 1;
  for (int i = 0; i < 7; i++) inv *= 2 - p[0] * inv; /* Newton: p^-1 mod 2^64 */
  f->inv = (uint64_t)(0 - inv);
  /* R mod p and R^2 mod p by repeated doubling of 1 */
  fe acc = {{1, 0, 0, 0}};
  for (int i = 0; i < 512; i++) {
    fe_add(f, &acc, &acc, &acc);
    if (i == 255) f->one = acc;
  }
  f->r2 = acc;
  uint64_t five[4] = {5, 0, 0, 0};
  fe_from_canon(f, &f->b, five);
  /* p - 1 = 2^32 * t */
  uint64_t pm1[4] = {p[0] - 1, p[1], p[2], p[3]};
  for (int i = 0; i < 4; i++) f->t[i] = (pm1[i] >> 32) | (i < 3 ? pm1[i + 1] << 32 : 0);
  /* non-residue: smallest g with g^((p-1)/2) = -1 */
  uint64_t half[4];
  for (int i = 0; i < 4; i++) half[i] = (pm1[i] >> 1) | (i < 3 ? pm1[i + 1] << 63 : 0);
  fe minus1;
  fe_neg(f, &minus1, &f->one);
  for (uint64_t g = 2;; g++) {
    uint64_t gc[4] = {g, 0, 0, 0};
    fe gm, e;
    fe_from_canon(f, &gm, gc);
    fe_pow(f, &e, &gm, half);
    if (fe_eq(&e, &minus1)) {
      fe_pow(f, &f->root_of_unity, &gm, f->t);
      break;
    }
  }
}

/* ---- curve (Jacobian, a = 0, b = 5) ------------------------------------------------ */
/* formulas as in ark-ec 0.3 short-Weierstrass Jacobian: dbl-2009-l, add-2007-bl */
static void jac_set_inf(const field* f, jac* r) {
  memset(r, 0, sizeof *r);
  r->y = f->one; /* ark's zero() = (0, 1, 0) */
}
static int jac_is_inf(const jac* p) { return fe_is_zero(&p->z); }
static void jac_dbl(const field* f, jac* r, const jac* p) {
  if (jac_is_inf(p)) { *r = *p; return; }
  fe a, b, c, d, e, ff, t, x3, y3, z3;
  fe_sqr(f, &a, &p->x);
  fe_sqr(f, &b, &p->y);
  fe_sqr(f, &c, &b);
  fe_add(f, &t, &p->x, &b);
  fe_sqr(f, &t, &t);
  fe_sub(f, &t, &t, &a);
  fe_sub(f, &t, &t, &c);
  fe_dbl(f, &d, &t);
  fe_dbl(f, &e, &a);
  fe_add(f, &e, &e, &a);
  fe_sqr(f, &ff, &e);
  fe_mul(f, &z3, &p->y, &p->z);
  fe_dbl(f, &z3, &z3);
  fe_sub(f, &x3, &ff, &d);
  fe_sub(f, &x3, &x3, &d);
  fe_sub(f, &t, &d, &x3);
  fe_mul(f, &y3, &e, &t);
  fe_dbl(f, &c, &c);
  fe_dbl(f, &c, &c);
  fe_dbl(f, &c, &c);
  fe_sub(f, &y3, &y3, &c);
  r->x = x3; r->y = y3; r->z = z3;
}
static void jac_add(const field* f, jac* r, const jac* p, const jac* q) {
  if (jac_is_inf(p)) { *r = *q; return; }
  if (jac_is_inf(q)) { *r = *p; return; }
  fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
  fe_sqr(f, &z1z1, &p->z);
  fe_sqr(f, &z2z2, &q->z);
  fe_mul(f, &u1, &p->x, &z2z2);
  fe_mul(f, &u2, &q->x, &z1z1);
  fe_mul(f, &s1, &p->y, &q->z);
  fe_mul(f, &s1, &s1, &z2z2);
  fe_mul(f, &s2, &q->y, &p->z);
  fe_mul(f, &s2, &s2, &z1z1);
  if (fe_eq(&u1, &u2)) {
    if (fe_eq(&s1, &s2)) { jac_dbl(f, r, p); return; }
    jac_set_inf(f, r);
    return;
  }
  fe_sub(f, &h, &u2, &u1);
  fe_dbl(f, &i, &h);
  fe_sqr(f, &i, &i);
  fe_mul(f, &j, &h, &i);
  fe_sub(f, &rr, &s2, &s1);
  fe_dbl(f, &rr, &rr);
  fe_mul(f, &v, &u1, &i);
  fe_sqr(f, &x3, &rr);
  fe_sub(f, &x3, &x3, &j);
  fe_sub(f, &x3, &x3, &v);
  fe_sub(f, &x3, &x3, &v);
  fe_sub(f, &t, &v, &x3);
  fe_mul(f, &y3, &rr, &t);
  fe_mul(f, &t, &s1, &j);
  fe_dbl(f, &t, &t);
  fe_sub(f, &y3, &y3, &t);
  fe_add(f, &z3, &p->z, &q->z);
  fe_sqr(f, &z3, &z3);
  fe_sub(f, &z3, &z3, &z1z1);
  fe_sub(f, &z3, &z3, &z2z2);
  fe_mul(f, &z3, &z3, &h);
  r->x = x3; r->y = y3; r->z = z3;
}
/* Group::mul: MSB-first double-and-add over the canonical scalar (msm.rs:21, bp_iter.rs:99) */
static void jac_mul(const field* f, jac* r, const jac* p, const uint64_t k[4]) {
  jac acc;
  jac_set_inf(f, &acc);
  int started = 0;
  for (int i = 3; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      if (started) jac_dbl(f, &acc, &acc);
      if ((k[i] >> b) & 1) { jac_add(f, &acc, &acc, p); started = 1; }
    }
  *r = acc;
}
/* normalise: (X/Z^2, Y/Z^3, 1); identity -> (0, 1, 0) */
static void jac_normalise(const field* f, jac* r, const jac* p) {
  if (jac_is_inf(p)) { jac_set_inf(f, r); return; }
  fe zi, zi2, zi3;
  fe_inv(f, &zi, &p->z);
  fe_sqr(f, &zi2, &zi);
  fe_mul(f, &zi3, &zi2, &zi);
  fe_mul(f, &r->x, &p->x, &zi2);
  fe_mul(f, &r->y, &p->y, &zi3);
  r->z = f->one;
}

/* ---- threads --------------------------------------------------------------------- */
typedef void (*job_fn)(void* ctx, int idx);
typedef struct { job_fn fn; void* ctx; int njobs; int next; pthread_mutex_t mu; } pool;
static void* pool_worker(void* arg) {
  pool* pl = (pool*)arg;
  for (;;) {
    pthread_mutex_lock(&pl->mu);
    int i = pl->next++;
    pthread_mutex_unlock(&pl->mu);
    if (i >= pl->njobs) return NULL;
    pl->fn(pl->ctx, i);
  }
}
static void run_jobs(job_fn fn, void* ctx, int njobs) {
  int nt = g_threads < njobs ? g_threads : njobs;
  if (nt <= 1) { for (int i = 0; i < njobs; i++) fn(ctx, i); return; }
  pool pl = {fn, ctx, njobs, 0, PTHREAD_MUTEX_INITIALIZER};
  pthread_t th[256];
  if (nt > 256) nt = 256;
  for (int i = 0; i < nt; i++) pthread_create(&th[i], NULL, pool_worker, &pl);
  for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
}

/* ---- MSM ------------------------------------------------------------------------- */
static int ark_log2(size_t x) { /* ceil(log2 x), 0 for x <= 1 */
  int l = 0;
  if (x <= 1) return 0;
  x -= 1;
  while (x) { l++; x >>= 1; }
  return l;
}
static int pippenger_window(size_t n) { return n < 32 ? 3 : ark_log2(n) * 69 / 100 + 2; } /* msm.rs:83-86,102-106 */

typedef struct {
  const field* fq;
  const jac* bases;
  const uint64_t (*sc)[4]; /* canonical scalars */
  size_t n;
  int c;
  jac* window_sums;
} pip_ctx;

static void pip_window(void* vctx, int widx) {
  pip_ctx* x = (pip_ctx*)vctx;
  const field* f = x->fq;
  int c = x->c;
  size_t w_start = (size_t)widx * c;
  size_t nb = ((size_t)1 << c) - 1;
  jac res;
  jac_set_inf(f, &res);
  jac* buckets = (jac*)malloc(nb * sizeof(jac));
  for (size_t i = 0; i < nb; i++) jac_set_inf(f, &buckets[i]);
  for (size_t i = 0; i < x->n; i++) {
    const uint64_t* s = x->sc[i];
    if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;                 /* msm.rs:100 */
    if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) {                   /* msm.rs:127-132 */
      if (w_start == 0) jac_add(f, &res, &res, &x->bases[i]);
      continue;
    }
    /* scalar.divn(w_start); scalar % 2^c  (msm.rs:136-141) */
    size_t wi = w_start >> 6, sh = w_start & 63;
    uint64_t d = s[wi] >> sh;
    if (sh && wi + 1 < 4) d |= s[wi + 1] << (64 - sh);
    d &= ((uint64_t)1 << c) - 1;
    if (d != 0) jac_add(f, &buckets[d - 1], &buckets[d - 1], &x->bases[i]); /* msm.rs:148 */
  }
  jac running;
  jac_set_inf(f, &running);
  for (size_t i = nb; i-- > 0;) {                                    /* msm.rs:167-171 */
    jac_add(f, &running, &running, &buckets[i]);
    jac_add(f, &res, &res, &running);
  }
  free(buckets);
  x->window_sums[widx] = res;
}

static void msm_pippenger_impl(int curve, const jac* bases, const uint64_t (*sc_mont)[4], size_t n, jac* out) {
  const field* fq = &F[curve];
  const field* fr = &F[1 - curve];
  uint64_t (*sc)[4] = (uint64_t(*)[4])malloc((n ? n : 1) * 32);
  for (size_t i = 0; i < n; i++) {
    fe t;
    memcpy(t.v, sc_mont[i], 32);
    fe_to_canon(fr, sc[i], &t);                                      /* into_repr, msm.rs:96 */
  }
  int c = pippenger_window(n);
  int nwin = (255 + c - 1) / c;                                      /* window starts 0, c, .. < 255 */
  jac* ws = (jac*)malloc(nwin * sizeof(jac));
  pip_ctx ctx = {fq, bases, (const uint64_t(*)[4])sc, n, c, ws};
  run_jobs(pip_window, &ctx, nwin);
  jac lowest = ws[0], total;
  jac_set_inf(fq, &total);
  for (int i = nwin - 1; i >= 1; i--) {                              /* msm.rs:177-190 */
    jac_add(fq, &total, &total, &ws[i]);
    for (int k = 0; k < c; k++) jac_dbl(fq, &total, &total);
  }
  jac_add(fq, out, &lowest, &total);
  free(ws);
  free(sc);
}

static void msm_naive_impl(int curve, const jac* bases, const uint64_t (*sc_mont)[4], size_t n, jac* out) {
  const field* fq = &F[curve];
  const field* fr = &F[1 - curve];
  jac acc;
  jac_set_inf(fq, &acc);
  for (size_t i = 0; i < n; i++) {
    fe t;
    uint64_t k[4];
    memcpy(t.v, sc_mont[i], 32);
    fe_to_canon(fr, k, &t);
    jac m;
    jac_mul(fq, &m, &bases[i], k);
    jac_add(fq, &acc, &acc, &m);
  }
  *out = acc;
}

/* ---- ChaCha / BlockRng / BLAKE2s / FiatShamirRng ----------------------------------- */
#define ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define QR(a, b, c, d) \
  a += b; d ^= a; d = ROTL32(d, 16); c += d; b ^= c; b = ROTL32(b, 12); \
  a += b; d ^= a; d = ROTL32(d, 8);  c += d; b ^= c; b = ROTL32(b, 7);
static void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
  uint32_t st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
  memcpy(st + 4, key, 32);
  st[12] = (uint32_t)counter; st[13] = (uint32_t)(counter >> 32);
  st[14] = (uint32_t)stream;  st[15] = (uint32_t)(stream >> 32);
  uint32_t x[16];
  memcpy(x, st, 64);
  for (int i = 0; i < rounds / 2; i++) {
    QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
    QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
  }
  for (int i = 0; i < 16; i++) out[i] = x[i] + st[i];
}
/* rand_core BlockRng over rand_chacha: 64-word buffer = 4 blocks per refill */
typedef struct { uint32_t key[8]; int rounds; uint64_t counter; uint32_t buf[64]; int index; } rng_t;
static void rng_init(rng_t* r, const uint8_t seed[32], int rounds) {
  memcpy(r->key, seed, 32); /* little-endian host */
  r->rounds = rounds; r->counter = 0; r->index = 64;
}
static void rng_refill(rng_t* r) {
  for (int i = 0; i < 4; i++) chacha_block(r->key, r->counter + i, 0, r->rounds, r->buf + 16 * i);
  r->counter += 4;
}
static uint32_t rng_u32(rng_t* r) {
  if (r->index >= 64) { rng_refill(r); r->index = 0; }
  return r->buf[r->index++];
}
static uint64_t rng_u64(rng_t* r) {
  uint64_t lo, hi;
  if (r->index < 63) { lo = r->buf[r->index]; hi = r->buf[r->index + 1]; r->index += 2; }
  else if (r->index >= 64) { rng_refill(r); lo = r->buf[0]; hi = r->buf[1]; r->index = 2; }
  else { lo = r->buf[63]; rng_refill(r); hi = r->buf[0]; r->index = 1; }
  return lo | (hi << 32);
}
/* ark-ff Fp256::rand: 4 x next_u64, clear the top bit, accept iff < p; the limbs are the Montgomery repr */
static void rng_field(rng_t* r, const field* f, fe* out) {
  for (;;) {
    for (int i = 0; i < 4; i++) out->v[i] = rng_u64(r);
    out->v[3] &= 0x7fffffffffffffffull;
    if (!ge4(out->v, f->p)) return;
  }
}
static int canon_less(const field* f, const fe* a, const fe* b) {
  uint64_t ca[4], cb[4];
  fe_to_canon(f, ca, a);
  fe_to_canon(f, cb, b);
  return !ge4(ca, cb);
}
/* ark-ec SW GroupProjective::rand (cofactor 1 -> Z = 1) */
static void rng_point(rng_t* r, const field* f, jac* out) {
  for (;;) {
    fe x, t, y, negy;
    rng_field(r, f, &x);
    int greatest = (int)(rng_u32(r) >> 31);
    fe_sqr(f, &t, &x);
    fe_mul(f, &t, &t, &x);
    fe_add(f, &t, &t, &f->b);
    if (!fe_sqrt(f, &y, &t)) continue;
    fe_neg(f, &negy, &y);
    int take_y = canon_less(f, &y, &negy) ^ greatest;
    out->x = x;
    out->y = take_y ? y : negy;
    out->z = f->one;
    return;
  }
}

static const uint32_t B2S_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
#define ROTR32(v, n) (((v) >> (n)) | ((v) << (32 - (n))))
static void b2s_compress(uint32_t h[8], const uint8_t block[64], uint64_t t, int last) {
  uint32_t m[16], v[16];
  memcpy(m, block, 64);
  for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2S_IV[i]; }
  v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
  if (last) v[14] = ~v[14];
#define G(a, b, c, d, x, y) \
  v[a] += v[b] + (x); v[d] = ROTR32(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ROTR32(v[b] ^ v[c], 12); \
  v[a] += v[b] + (y); v[d] = ROTR32(v[d] ^ v[a], 8);  v[c] += v[d]; v[b] = ROTR32(v[b] ^ v[c], 7);
  for (int r = 0; r < 10; r++) {
    const uint8_t* s = B2S_SIGMA[r];
    G(0, 4, 8, 12, m[s[0]], m[s[1]]) G(1, 5, 9, 13, m[s[2]], m[s[3]]) G(2, 6, 10, 14, m[s[4]], m[s[5]]) G(3, 7, 11, 15, m[s[6]], m[s[7]])
    G(0, 5, 10, 15, m[s[8]], m[s[9]]) G(1, 6, 11, 12, m[s[10]], m[s[11]]) G(2, 7, 8, 13, m[s[12]], m[s[13]]) G(3, 4, 9, 14, m[s[14]], m[s[15]])
  }
  for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}
static void blake2s256(const uint8_t* in, size_t len, uint8_t out[32]) {
  uint32_t h[8];
  memcpy(h, B2S_IV, 32);
  h[0] ^= 0x01010020; /* digest 32 bytes, no key */
  uint64_t t = 0;
  uint8_t block[64];
  while (len > 64) {
    t += 64;
    b2s_compress(h, in, t, 0);
    in += 64; len -= 64;
  }
  memset(block, 0, 64);
  memcpy(block, in, len);
  t += len;
  b2s_compress(h, block, t, 1);
  memcpy(out, h, 32);
}
/* ark-marlin 0.3 FiatShamirRng<Blake2s>: seed = H(bytes); absorb: seed = H(new || seed); r = ChaCha20(seed) */
typedef struct { uint8_t seed[32]; rng_t r; } fs_t;
static void fs_init(fs_t* fs, const uint8_t* bytes, size_t len) {
  blake2s256(bytes, len, fs->seed);
  rng_init(&fs->r, fs->seed, 20);
}
static void fs_absorb(fs_t* fs, const uint8_t* bytes, size_t len) {
  uint8_t buf[512];
  memcpy(buf, bytes, len);
  memcpy(buf + len, fs->seed, 32);
  blake2s256(buf, len + 32, fs->seed);
  rng_init(&fs->r, fs->seed, 20);
}
/* ToBytes for GroupProjective after the normalise-at-boundary rule: canonical x | y | 1, identity (0, 1, 0) */
static void point_bytes(const field* f, const jac* p_norm, uint8_t out[96]) {
  uint64_t c[4];
  fe_to_canon(f, c, &p_norm->x); memcpy(out, c, 32);
  fe_to_canon(f, c, &p_norm->y); memcpy(out + 32, c, 32);
  fe_to_canon(f, c, &p_norm->z); memcpy(out + 64, c, 32);
}

/* ---- folds / ip ------------------------------------------------------------------ */
typedef struct { const field* f; const fe* v; fe* out; size_t n; fe s_lo, s_hi; size_t chunk; } sfold_ctx;
static void sfold_job(void* vc, int idx) {
  sfold_ctx* c = (sfold_ctx*)vc;
  size_t lo = (size_t)idx * c->chunk, hi = lo + c->chunk;
  if (hi > c->n) hi = c->n;
  for (size_t i = lo; i < hi; i++) {
    fe t0, t1;
    fe_mul(c->f, &t0, &c->v[i], &c->s_lo);
    fe_mul(c->f, &t1, &c->v[i + c->n], &c->s_hi);
    fe_add(c->f, &c->out[i], &t0, &t1);
  }
}
typedef struct { const field* f; const jac* g; jac* out; size_t n; uint64_t k_lo[4], k_hi[4]; size_t chunk; } pfold_ctx;
static void pfold_job(void* vc, int idx) {
  pfold_ctx* c = (pfold_ctx*)vc;
  size_t lo = (size_t)idx * c->chunk, hi = lo + c->chunk;
  if (hi > c->n) hi = c->n;
  for (size_t i = lo; i < hi; i++) {
    jac a, b;
    jac_mul(c->f, &a, &c->g[i], c->k_lo);          /* l.mul(&x_inv) */
    jac_mul(c->f, &b, &c->g[i + c->n], c->k_hi);   /* r.mul(&x) */
    jac_add(c->f, &c->out[i], &a, &b);
  }
}
static void fold_scalars_impl(int curve, const fe* v, size_t n, const fe* s_lo, const fe* s_hi, fe* out) {
  sfold_ctx c = {&F[1 - curve], v, out, n, *s_lo, *s_hi, 4096};
  run_jobs(sfold_job, &c, (int)((n + 4095) / 4096));
}
static void fold_points_impl(int curve, const jac* g, size_t n, const fe* s_lo, const fe* s_hi, jac* out) {
  pfold_ctx c = {&F[curve], g, out, n, {0}, {0}, 64};
  fe_to_canon(&F[1 - curve], c.k_lo, s_lo);
  fe_to_canon(&F[1 - curve], c.k_hi, s_hi);
  run_jobs(pfold_job, &c, (int)((n + 63) / 64));
}
static void ip_impl(int curve, const fe* a, const fe* b, size_t n, fe* out) {
  const field* f = &F[1 - curve];
  fe acc = {{0, 0, 0, 0}};
  for (size_t i = 0; i < n; i++) {
    fe t;
    fe_mul(f, &t, &a[i], &b[i]);
    fe_add(f, &acc, &acc, &t);
  }
  *out = acc;
}

/* ================================ exported API ===================================== */
int po_init(int threads) {
  if (!g_init) {
    field_init(&F[0], P0);
    field_init(&F[1], P1);
    g_init = 1;
  }
  g_threads = threads < 1 ? 1 : threads;
  return 0;
}
int po_pippenger_window(size_t n) { return pippenger_window(n); }

/* out_raw: the Jacobian triple the reference's sequence of additions yields ("faithful Z");
 * out_norm: (x, y, 1) / (0, 1, 0).  Either may be NULL. */
int po_msm_pippenger(int curve, const uint64_t* points_xyz, const uint64_t* scalars, size_t n, uint64_t* out_raw, uint64_t* out_norm) {
  jac r, nr;
  msm_pippenger_impl(curve, (const jac*)points_xyz, (const uint64_t(*)[4])scalars, n, &r);
  if (out_raw) memcpy(out_raw, &r, 96);
  if (out_norm) { jac_normalise(&F[curve], &nr, &r); memcpy(out_norm, &nr, 96); }
  return 0;
}
int po_msm_naive(int curve, const uint64_t* points_xyz, const uint64_t* scalars, size_t n, uint64_t* out_raw, uint64_t* out_norm) {
  jac r, nr;
  msm_naive_impl(curve, (const jac*)points_xyz, (const uint64_t(*)[4])scalars, n, &r);
  if (out_raw) memcpy(out_raw, &r, 96);
  if (out_norm) { jac_normalise(&F[curve], &nr, &r); memcpy(out_norm, &nr, 96); }
  return 0;
}
int po_ip(int curve, const uint64_t* a, const uint64_t* b, size_t n, uint64_t out[4]) {
  ip_impl(curve, (const fe*)a, (const fe*)b, n, (fe*)out);
  return 0;
}
int po_fold_scalars(int curve, const uint64_t* v, size_t n_out, const uint64_t s_lo[4], const uint64_t s_hi[4], uint64_t* out) {
  fold_scalars_impl(curve, (const fe*)v, n_out, (const fe*)s_lo, (const fe*)s_hi, (fe*)out);
  return 0;
}
/* normalise != 0: outputs (x, y, 1) / (0, 1, 0) */
int po_fold_points(int curve, const uint64_t* g_xyz, size_t n_out, const uint64_t s_lo[4], const uint64_t s_hi[4], uint64_t* out_xyz, int normalise) {
  jac* out = (jac*)out_xyz;
  fold_points_impl(curve, (const jac*)g_xyz, n_out, (const fe*)s_lo, (const fe*)s_hi, out);
  if (normalise) for (size_t i = 0; i < n_out; i++) { jac t; jac_normalise(&F[curve], &t, &out[i]); out[i] = t; }
  return 0;
}
int po_normalise(int curve, const uint64_t* in_xyz, size_t n, uint64_t* out_xyz) {
  for (size_t i = 0; i < n; i++) { jac t; jac_normalise(&F[curve], &t, (const jac*)in_xyz + i); memcpy(out_xyz + 12 * i, &t, 96); }
  return 0;
}
int po_scalar_inverse(int curve, const uint64_t x[4], uint64_t out[4]) {
  fe_inv(&F[1 - curve], (fe*)out, (const fe*)x);
  return 0;
}
/* ---- vector utilities used by the k-ary protocols (src/util/mod.rs:25-85) ------------------------ */
/* sum_vecs(chunks.zip(y).map(|(c, y)| scale_vec(y, c)), ck)  (src/proofs/bp_rec_kary.rs:100-104,
 * src/reductions/ipa_to_bp_unroll.rs:248-252): out[i] = sum_j y[j] * v[j*ck + i], accumulated in
 * the order of the fold (zero + term 0 + term 1 ...). */
int po_fold_scalars_kary(int curve, const uint64_t* v, size_t ck, size_t k, const uint64_t* y, uint64_t* out) {
  const field* f = &F[1 - curve];
  const fe* vv = (const fe*)v;
  const fe* yy = (const fe*)y;
  for (size_t i = 0; i < ck; i++) {
    fe acc = {{0, 0, 0, 0}};
    for (size_t j = 0; j < k; j++) {
      fe t;
      fe_mul(f, &t, &vv[j * ck + i], &yy[j]);
      fe_add(f, &acc, &acc, &t);
    }
    ((fe*)out)[i] = acc;
  }
  return 0;
}
/* same for generators (bp_rec_kary.rs:105-112, ipa_to_bp_unroll.rs:253-260): scale_vec on a
 * Vec<G> is `c *= s`, i.e. Group::mul; outputs normalised. */
typedef struct { const field* f; const jac* g; jac* out; size_t ck, k; const uint64_t (*yc)[4]; } kfold_ctx;
static void kfold_job(void* vc, int idx) {
  kfold_ctx* c = (kfold_ctx*)vc;
  size_t lo = (size_t)idx * 16, hi = lo + 16;
  if (hi > c->ck) hi = c->ck;
  for (size_t i = lo; i < hi; i++) {
    jac acc, t, u;
    jac_set_inf(c->f, &acc);
    for (size_t j = 0; j < c->k; j++) {
      jac_mul(c->f, &t, &c->g[j * c->ck + i], c->yc[j]);
      jac_add(c->f, &u, &acc, &t);
      acc = u;
    }
    jac_normalise(c->f, &c->out[i], &acc);
  }
}
int po_fold_points_kary(int curve, const uint64_t* g_xyz, size_t ck, size_t k, const uint64_t* y, uint64_t* out_xyz) {
  uint64_t (*yc)[4] = malloc((k ? k : 1) * 32);
  for (size_t j = 0; j < k; j++) fe_to_canon(&F[1 - curve], yc[j], (const fe*)y + j);
  kfold_ctx c = {&F[curve], (const jac*)g_xyz, (jac*)out_xyz, ck, k, (const uint64_t(*)[4])yc};
  run_jobs(kfold_job, &c, (int)((ck + 15) / 16));
  free(yc);
  return 0;
}
/* hadamard, src/util/mod.rs:80-85 */
int po_hadamard(int curve, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  for (size_t i = 0; i < n; i++) fe_mul(&F[1 - curve], (fe*)out + i, (const fe*)a + i, (const fe*)b + i);
  return 0;
}
/* hadamard_gp, src/util/mod.rs:70-78: out[i] = bases[i].mul(scalars[i]); outputs normalised */
typedef struct { const field* f; const field* fs; const jac* g; const fe* s; jac* out; size_t n; } hgp_ctx;
static void hgp_job(void* vc, int idx) {
  hgp_ctx* c = (hgp_ctx*)vc;
  size_t lo = (size_t)idx * 16, hi = lo + 16;
  if (hi > c->n) hi = c->n;
  for (size_t i = lo; i < hi; i++) {
    uint64_t k[4];
    jac t;
    fe_to_canon(c->fs, k, &c->s[i]);
    jac_mul(c->f, &t, &c->g[i], k);
    jac_normalise(c->f, &c->out[i], &t);
  }
}
int po_hadamard_points(int curve, const uint64_t* bases_xyz, const uint64_t* scalars, size_t n, uint64_t* out_xyz) {
  hgp_ctx c = {&F[curve], &F[1 - curve], (const jac*)bases_xyz, (const fe*)scalars, (jac*)out_xyz, n};
  run_jobs(hgp_job, &c, (int)((n + 15) / 16));
  return 0;
}
/* powers, src/util/mod.rs:57-64: first = f.pow([start]); successors multiply by f */
int po_powers(int curve, const uint64_t f_in[4], uint64_t start, size_t n, uint64_t* out) {
  const field* f = &F[1 - curve];
  const uint64_t e[4] = {start, 0, 0, 0};
  fe cur;
  fe_pow(f, &cur, (const fe*)f_in, e);
  for (size_t i = 0; i < n; i++) {
    ((fe*)out)[i] = cur;
    fe nx;
    fe_mul(f, &nx, &cur, (const fe*)f_in);
    cur = nx;
  }
  return 0;
}
/* mat_vec_mult, src/relations/com_r1cs.rs:43-57: result[r] += mat_val * vec[c] over the rows' (value, column)
 * pairs; the matrix comes as compressed rows (row r = entries row_ptr[r] .. row_ptr[r+1]).  Returns -1 where
 * the reference asserts (c < vec.len()). */
int po_mat_vec_mult(int curve, const uint64_t* row_ptr, const uint64_t* col_idx, const uint64_t* vals, size_t n_rows,
                    const uint64_t* vec, size_t vec_len, uint64_t* out) {
  const field* f = &F[1 - curve];
  for (size_t r = 0; r < n_rows; r++) {
    fe acc;
    memset(&acc, 0, sizeof acc);
    for (uint64_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
      if (col_idx[e] >= vec_len) return -1;
      fe t;
      fe_mul(f, &t, (const fe*)vals + e, (const fe*)vec + col_idx[e]);
      fe_add(f, &acc, &acc, &t);
    }
    ((fe*)out)[r] = acc;
  }
  return 0;
}
/* vec_mat_mult, src/relations/com_r1cs.rs:67-83: result[c] += mat_val * vec[r]; -1 where the reference asserts
 * (mat_cols(mat) <= output_size, vec.len() == mat.len()) */
int po_vec_mat_mult(int curve, const uint64_t* vec, size_t vec_len, const uint64_t* row_ptr, const uint64_t* col_idx,
                    const uint64_t* vals, size_t n_rows, size_t output_size, uint64_t* out) {
  const field* f = &F[1 - curve];
  if (vec_len != n_rows) return -1;
  for (size_t r = 0; r < n_rows; r++)
    for (uint64_t e = row_ptr[r]; e < row_ptr[r + 1]; e++)
      if (col_idx[e] >= output_size) return -1;
  memset(out, 0, output_size * 32);
  for (size_t r = 0; r < n_rows; r++)
    for (uint64_t e = row_ptr[r]; e < row_ptr[r + 1]; e++) {
      fe t;
      fe_mul(f, &t, (const fe*)vals + e, (const fe*)vec + r);
      fe_add(f, (fe*)out + col_idx[e], (fe*)out + col_idx[e], &t);
    }
  return 0;
}
/* Fr::from(u64) in ark memory form (used by the padding-helper known answers, util/mod.rs:146-203) */
int po_scalar_from_u64(int curve, uint64_t v, uint64_t out[4]) {
  const uint64_t c[4] = {v, 0, 0, 0};
  fe_from_canon(&F[1 - curve], (fe*)out, c);
  return 0;
}
/* zero_pad_to_multiple (util/mod.rs:103-111) and zero_intersperse_to_multiple (:117-134) on 32-byte
 * elements; return the output length (out may be NULL to query it). */
size_t po_zero_pad_to_multiple(const uint64_t* in, size_t n, size_t k, uint64_t* out) {
  size_t chunk = (n - 1) / k + 1, want = chunk * k;
  if (out) {
    memcpy(out, in, n * 32);
    if (want > n) memset(out + 4 * n, 0, (want - n) * 32);
  }
  return want > n ? want : n;
}
size_t po_zero_intersperse_to_multiple(const uint64_t* in, size_t n, size_t k, uint64_t* out) {
  size_t block = (n - 1) / k + 1, fin = block * k, nshort = fin - n, o = 0;
  if (!out) return fin;
  for (size_t i = 0; i < k; i++) {
    if (i >= k - nshort) {
      size_t start = block * (k - nshort) + (block - 1) * (i + nshort - k);
      memcpy(out + 4 * o, in + 4 * start, (block - 1) * 32);
      o += block - 1;
      memset(out + 4 * o, 0, 32);
      o += 1;
    } else {
      memcpy(out + 4 * o, in + 4 * block * i, block * 32);
      o += block;
    }
  }
  return o;
}
/* on-curve check of a Jacobian point: Y^2 = X^3 + 5 Z^6 */
int po_on_curve(int curve, const uint64_t* p_xyz) {
  const field* f = &F[curve];
  const jac* p = (const jac*)p_xyz;
  if (jac_is_inf(p)) return 1;
  fe l, r, z2, z6;
  fe_sqr(f, &l, &p->y);
  fe_sqr(f, &r, &p->x); fe_mul(f, &r, &r, &p->x);
  fe_sqr(f, &z2, &p->z); fe_sqr(f, &z6, &z2); fe_mul(f, &z6, &z6, &z2);
  fe_mul(f, &z6, &z6, &f->b);
  fe_add(f, &r, &r, &z6);
  return fe_eq(&l, &r);
}

/* examples/msm_bench.rs:40-44: one ark_std::test_rng() stream, n points then n scalars */
static const uint8_t TEST_RNG_SEED[32] = {1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0};
int po_msm_bench_inputs(int curve, size_t n, uint64_t* points_xyz, uint64_t* scalars) {
  rng_t r;
  rng_init(&r, TEST_RNG_SEED, 12);
  for (size_t i = 0; i < n; i++) rng_point(&r, &F[curve], (jac*)points_xyz + i);
  for (size_t i = 0; i < n; i++) rng_field(&r, &F[1 - curve], (fe*)scalars + i);
  return 0;
}
/* src/lib.rs:111-118 + src/relations/ipa.rs:28-36,90-97,110-115: a_gens, b_gens, ip_gen, a, b, fs seed */
int po_ipa_inputs(int curve, size_t n, uint64_t* a_gens, uint64_t* b_gens, uint64_t* ip_gen, uint64_t* a, uint64_t* b, uint8_t fs_seed[32]) {
  rng_t r;
  rng_init(&r, TEST_RNG_SEED, 12);
  for (size_t i = 0; i < n; i++) rng_point(&r, &F[curve], (jac*)a_gens + i);
  for (size_t i = 0; i < n; i++) rng_point(&r, &F[curve], (jac*)b_gens + i);
  rng_point(&r, &F[curve], (jac*)ip_gen);
  for (size_t i = 0; i < n; i++) rng_field(&r, &F[1 - curve], (fe*)a + i);
  for (size_t i = 0; i < n; i++) rng_field(&r, &F[1 - curve], (fe*)b + i);
  for (int i = 0; i < 32; i++) fs_seed[i] = (uint8_t)rng_u32(&r); /* rng.gen::<[u8;32]>() */
  return 0;
}
/* Seeded bulk generator for large synthetic inputs (NOT the reference recipe): point i comes
 * from its own ChaCha12 stream keyed by (seed, i), so generation parallelises. */
typedef struct { int curve; uint64_t seed; jac* pts; size_t n; size_t chunk; } gen_ctx;
static void gen_job(void* vc, int idx) {
  gen_ctx* c = (gen_ctx*)vc;
  size_t lo = (size_t)idx * c->chunk, hi = lo + c->chunk;
  if (hi > c->n) hi = c->n;
  for (size_t i = lo; i < hi; i++) {
    uint8_t seed[32] = {0};
    memcpy(seed, &c->seed, 8);
    memcpy(seed + 8, &i, 8);
    rng_t r;
    rng_init(&r, seed, 12);
    rng_point(&r, &F[c->curve], &c->pts[i]);
  }
}
int po_gen_points(int curve, uint64_t seed, size_t n, uint64_t* points_xyz) {
  gen_ctx c = {curve, seed, (jac*)points_xyz, n, 256};
  run_jobs(gen_job, &c, (int)((n + 255) / 256));
  return 0;
}
int po_gen_scalars(int curve, uint64_t seed, size_t n, uint64_t* scalars) {
  uint8_t s[32] = {0};
  memcpy(s, &seed, 8);
  s[31] = 1;
  rng_t r;
  rng_init(&r, s, 12);
  for (size_t i = 0; i < n; i++) rng_field(&r, &F[1 - curve], (fe*)scalars + i);
  return 0;
}

/* Bp::prove, src/proofs/bp_iter.rs:49-128, with every msm result normalised before it is absorbed.
 * n must be a power of two (the caller pads, bp_iter.rs:59-62).  ls/rs: log2(n) normalised points;
 * challenges: log2(n) scalars (ark form).  Inputs are consumed (folded in place on private copies). */
int po_bp_prove(int curve, const uint64_t* a_gens, const uint64_t* b_gens, const uint64_t* ip_gen, const uint64_t* a_in,
                const uint64_t* b_in, size_t n, const uint8_t fs_seed[32], uint64_t* ls, uint64_t* rs, uint64_t* challenges,
                uint64_t final_a[4], uint64_t final_b[4]) {
  const field* fq = &F[curve];
  const field* fr = &F[1 - curve];
  if (n == 0 || (n & (n - 1))) return -1;
  jac* A = (jac*)malloc(n * sizeof(jac));
  jac* B = (jac*)malloc(n * sizeof(jac));
  fe* a = (fe*)malloc(n * sizeof(fe));
  fe* b = (fe*)malloc(n * sizeof(fe));
  jac* mp = (jac*)malloc((n + 1) * sizeof(jac));
  fe* ms = (fe*)malloc((n + 1) * sizeof(fe));
  jac* tmpA = (jac*)malloc((n / 2 + 1) * sizeof(jac));
  fe* tmpa = (fe*)malloc((n / 2 + 1) * sizeof(fe));
  memcpy(A, a_gens, n * sizeof(jac));
  memcpy(B, b_gens, n * sizeof(jac));
  memcpy(a, a_in, n * sizeof(fe));
  memcpy(b, b_in, n * sizeof(fe));
  fs_t fs;
  fs_init(&fs, fs_seed, 32);
  size_t len = n;
  int round = 0;
  while (len > 1) {
    size_t h = len / 2;
    jac l, r, ln, rn;
    /* l = msm(A[h..] | B[..h] | Q, a[..h] | b[h..] | <a[..h], b[h..]>) */
    memcpy(mp, A + h, h * sizeof(jac)); memcpy(mp + h, B, h * sizeof(jac)); memcpy(mp + 2 * h, ip_gen, sizeof(jac));
    memcpy(ms, a, h * sizeof(fe)); memcpy(ms + h, b + h, h * sizeof(fe)); ip_impl(curve, a, b + h, h, ms + 2 * h);
    msm_pippenger_impl(curve, mp, (const uint64_t(*)[4])ms, 2 * h + 1, &l);
    /* r = msm(A[..h] | B[h..] | Q, a[h..] | b[..h] | <a[h..], b[..h]>) */
    memcpy(mp, A, h * sizeof(jac)); memcpy(mp + h, B + h, h * sizeof(jac)); memcpy(mp + 2 * h, ip_gen, sizeof(jac));
    memcpy(ms, a + h, h * sizeof(fe)); memcpy(ms + h, b, h * sizeof(fe)); ip_impl(curve, a + h, b, h, ms + 2 * h);
    msm_pippenger_impl(curve, mp, (const uint64_t(*)[4])ms, 2 * h + 1, &r);
    jac_normalise(fq, &ln, &l);
    jac_normalise(fq, &rn, &r);
    memcpy(ls + 12 * round, &ln, 96);
    memcpy(rs + 12 * round, &rn, 96);
    uint8_t buf[96];
    point_bytes(fq, &ln, buf); fs_absorb(&fs, buf, 96);
    point_bytes(fq, &rn, buf); fs_absorb(&fs, buf, 96);
    fe x, xi;
    rng_field(&fs.r, fr, &x);
    fe_inv(fr, &xi, &x);
    memcpy(challenges + 4 * round, &x, 32);
    fold_scalars_impl(curve, a, h, &x, &xi, tmpa); memcpy(a, tmpa, h * sizeof(fe));
    fold_scalars_impl(curve, b, h, &xi, &x, tmpa); memcpy(b, tmpa, h * sizeof(fe));
    fold_points_impl(curve, A, h, &xi, &x, tmpA); memcpy(A, tmpA, h * sizeof(jac));
    fold_points_impl(curve, B, h, &x, &xi, tmpA); memcpy(B, tmpA, h * sizeof(jac));
    len = h;
    round++;
  }
  memcpy(final_a, a, 32);
  memcpy(final_b, b, 32);
  free(A); free(B); free(a); free(b); free(mp); free(ms); free(tmpA); free(tmpa);
  return 0;
}

/* primitives exposed so tests can pin them against RFC vectors / hashlib */
int po_chacha_block(const uint8_t key[32], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
  uint32_t k[8];
  memcpy(k, key, 32);
  chacha_block(k, counter, stream, rounds, out);
  return 0;
}
int po_blake2s(const uint8_t* in, size_t len, uint8_t out[32]) { blake2s256(in, len, out); return 0; }
int po_fs_challenge(int curve, const uint8_t fs_seed[32], const uint8_t* absorb, size_t nabsorb96, uint64_t out[4]) {
  fs_t fs;
  fs_init(&fs, fs_seed, 32);
  for (size_t i = 0; i < nabsorb96; i++) fs_absorb(&fs, absorb + 96 * i, 96);
  rng_field(&fs.r, &F[1 - curve], (fe*)out);
  return 0;
}
int po_field_op(int fid, int op, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
  const field* f = &F[fid];
  switch (op) {
    case 0: fe_mul(f, (fe*)out, (const fe*)a, (const fe*)b); return 0;
    case 1: fe_add(f, (fe*)out, (const fe*)a, (const fe*)b); return 0;
    case 2: fe_sub(f, (fe*)out, (const fe*)a, (const fe*)b); return 0;
    case 3: fe_inv(f, (fe*)out, (const fe*)a); return 0;
    case 4: return fe_sqrt(f, (fe*)out, (const fe*)a) ? 0 : 1;
    default: return -1;
  }
}
