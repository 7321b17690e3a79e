/* bp_ref.c - plain C restatement of the reference's CPU algorithm for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ as a second checker and by bench.py's
 * cpu_baseline leg as the timed CPU baseline ("port").  The product (bulletproofs_amd/,
 * libbpgpu.so) never links or calls this file.
 *
 * It follows the reference's evaluation strategy literally (paths under /root/reference):
 *   mul        : left-to-right double-and-add, one product per term     (Utils.hs:73)
 *   sumExps    : N independent products summed pairwise                 (Utils.hs:76-80)
 *   hs'        : nm products hs_i * y^-i, each with its own `recip`     (MultiRangeProof/Verifier.hs:85)
 *   computeLRCommitment, verifyTPoly, verifyLRCommitment                (RangeProof/Internal.hs:169-206,
 *                                                                        MultiRangeProof/Verifier.hs:44-88)
 *   mkChallenges, mkOtherExponents, IPP verifyProof                     (InnerProductProof/Verifier.hs:18-73)
 *   oracle / shamir* / gs / hs / h                                      (Utils.hs:14-47,145-174)
 * It is GENEROUS to the reference: products run in Jacobian coordinates with one
 * inversion at the end (the reference's affine ladder inverts at every step) and field
 * arithmetic is 4x64-bit limbs instead of GMP Integers.
 *
 * PARITY: pinned against the Python oracle (itself pinned against OpenSSL) and the golden
 * vectors in tests/test_oracle_c.py.  Transcript text format / sqrt branch are the same
 * unpinned switches as in bp_oracle.py (style, sqrt_branch).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;          /* 256-bit integer, little-endian limbs */
typedef struct { fe x, y; int inf; } pt;       /* affine point, inf = 1 for O */
typedef struct { fe X, Y, Z; } jac;            /* Jacobian, Z = 0 for O */

static const fe FP = {{0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL}};
static const fe FR = {{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
static const fe GX = {{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}};
static const fe GY = {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}};

/* ---------------- 256-bit helpers ---------------- */
static int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int fe_cmp(const fe* a, const fe* b) {
  for (int i = 3; i >= 0; i--) {
    if (a->v[i] > b->v[i]) return 1;
    if (a->v[i] < b->v[i]) return -1;
  }
  return 0;
}
static uint64_t fe_add_raw(fe* r, const fe* a, const fe* b) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; r->v[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t fe_sub_raw(fe* r, const fe* a, const fe* b) {
  uint64_t bw = 0;
  for (int i = 0; i < 4; i++) {
    u128 t = (u128)a->v[i] - b->v[i] - bw;
    r->v[i] = (uint64_t)t;
    bw = (uint64_t)(t >> 64) & 1;
  }
  return bw;
}
static fe fe_from_u64(uint64_t x) { fe r = {{x, 0, 0, 0}}; return r; }

/* generic modular arithmetic for a modulus m = 2^256 - c, c < 2^130 (both p and r) */
static fe mod_add(const fe* a, const fe* b, const fe* m) {
  fe r, t;
  uint64_t cy = fe_add_raw(&r, a, b);
  if (cy || fe_cmp(&r, m) >= 0) { fe_sub_raw(&t, &r, m); return t; }
  return r;
}
static fe mod_sub(const fe* a, const fe* b, const fe* m) {
  fe r, t;
  if (fe_sub_raw(&r, a, b)) { fe_add_raw(&t, &r, m); return t; }
  return r;
}
static fe mod_neg(const fe* a, const fe* m) { fe z = {{0, 0, 0, 0}}; return mod_sub(&z, a, m); }

/* 512-bit product, then repeated folding with 2^256 = c (mod m) */
static fe mod_mul(const fe* a, const fe* b, const fe* m) {
  uint64_t t[8] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->v[i] * b->v[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    t[i + 4] = (uint64_t)c;
  }
  fe c256, zero = {{0, 0, 0, 0}};
  fe_sub_raw(&c256, &zero, m);          /* c = 2^256 - m, at most 3 limbs (c < 2^130) */
  /* fold: value = lo + hi * c, until hi == 0 */
  for (;;) {
    if ((t[4] | t[5] | t[6] | t[7]) == 0) break;
    uint64_t n[8] = {t[0], t[1], t[2], t[3], 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 3; j++) { c += (u128)t[4 + i] * c256.v[j] + n[i + j]; n[i + j] = (uint64_t)c; c >>= 64; }
      for (int k = i + 3; c && k < 8; k++) { c += n[k]; n[k] = (uint64_t)c; c >>= 64; }
    }
    memcpy(t, n, sizeof n);
  }
  fe r = {{t[0], t[1], t[2], t[3]}}, s;
  while (fe_cmp(&r, m) >= 0) { fe_sub_raw(&s, &r, m); r = s; }
  return r;
}
static fe mod_pow(const fe* a, const fe* e, const fe* m) {
  fe r = fe_from_u64(1);
  for (int i = 255; i >= 0; i--) {
    r = mod_mul(&r, &r, m);
    if ((e->v[i >> 6] >> (i & 63)) & 1) r = mod_mul(&r, a, m);
  }
  return r;
}
/* `recip`: a^(m-2) */
static fe mod_inv(const fe* a, const fe* m) {
  fe e, two = fe_from_u64(2);
  fe_sub_raw(&e, m, &two);
  return mod_pow(a, &e, m);
}
#define FQ_MUL(a, b) mod_mul(a, b, &FP)
#define FQ_ADD(a, b) mod_add(a, b, &FP)
#define FQ_SUB(a, b) mod_sub(a, b, &FP)

/* ---------------- group law ---------------- */
static pt pt_inf(void) { pt p; memset(&p, 0, sizeof p); p.inf = 1; return p; }
static pt pt_gen(void) { pt p; p.x = GX; p.y = GY; p.inf = 0; return p; }
static int pt_eq(const pt* a, const pt* b) {
  if (a->inf || b->inf) return a->inf && b->inf;
  return fe_cmp(&a->x, &b->x) == 0 && fe_cmp(&a->y, &b->y) == 0;
}
static pt pt_neg(const pt* a) { pt r = *a; if (!a->inf) r.y = mod_neg(&a->y, &FP); return r; }

/* `<>` on affine points (total) */
static pt pt_add(const pt* a, const pt* b) {
  if (a->inf) return *b;
  if (b->inf) return *a;
  fe lam;
  if (fe_cmp(&a->x, &b->x) == 0) {
    fe s = FQ_ADD(&a->y, &b->y);
    if (fe_is_zero(&s)) return pt_inf();
    fe xx = FQ_MUL(&a->x, &a->x), three = fe_from_u64(3), num = FQ_MUL(&xx, &three);
    fe den = FQ_ADD(&a->y, &a->y), di = mod_inv(&den, &FP);
    lam = FQ_MUL(&num, &di);
  } else {
    fe num = FQ_SUB(&b->y, &a->y), den = FQ_SUB(&b->x, &a->x), di = mod_inv(&den, &FP);
    lam = FQ_MUL(&num, &di);
  }
  pt r;
  r.inf = 0;
  fe l2 = FQ_MUL(&lam, &lam), t = FQ_SUB(&l2, &a->x);
  r.x = FQ_SUB(&t, &b->x);
  fe d = FQ_SUB(&a->x, &r.x), ld = FQ_MUL(&lam, &d);
  r.y = FQ_SUB(&ld, &a->y);
  return r;
}
static jac jac_dbl(const jac* p) {
  jac r;
  if (fe_is_zero(&p->Z) || fe_is_zero(&p->Y)) { memset(&r, 0, sizeof r); r.Y = fe_from_u64(1); return r; }
  fe yy = FQ_MUL(&p->Y, &p->Y), xyy = FQ_MUL(&p->X, &yy), four = fe_from_u64(4), S = FQ_MUL(&xyy, &four);
  fe xx = FQ_MUL(&p->X, &p->X), three = fe_from_u64(3), M = FQ_MUL(&xx, &three);
  fe mm = FQ_MUL(&M, &M), s2 = FQ_ADD(&S, &S);
  r.X = FQ_SUB(&mm, &s2);
  fe yyyy = FQ_MUL(&yy, &yy), eight = fe_from_u64(8), e8 = FQ_MUL(&yyyy, &eight);
  fe d = FQ_SUB(&S, &r.X), md = FQ_MUL(&M, &d);
  r.Y = FQ_SUB(&md, &e8);
  fe yz = FQ_MUL(&p->Y, &p->Z);
  r.Z = FQ_ADD(&yz, &yz);
  return r;
}
static jac jac_add_affine(const jac* p, const pt* q) {
  jac r;
  if (q->inf) return *p;
  if (fe_is_zero(&p->Z)) { r.X = q->x; r.Y = q->y; r.Z = fe_from_u64(1); return r; }
  fe zz = FQ_MUL(&p->Z, &p->Z), U2 = FQ_MUL(&q->x, &zz), zzz = FQ_MUL(&zz, &p->Z), S2 = FQ_MUL(&q->y, &zzz);
  fe H = FQ_SUB(&U2, &p->X), R = FQ_SUB(&S2, &p->Y);
  if (fe_is_zero(&H)) {
    if (fe_is_zero(&R)) return jac_dbl(p);
    memset(&r, 0, sizeof r); r.Y = fe_from_u64(1); return r;
  }
  fe HH = FQ_MUL(&H, &H), HHH = FQ_MUL(&HH, &H), V = FQ_MUL(&p->X, &HH);
  fe rr = FQ_MUL(&R, &R), t = FQ_SUB(&rr, &HHH), v2 = FQ_ADD(&V, &V);
  r.X = FQ_SUB(&t, &v2);
  fe d = FQ_SUB(&V, &r.X), rd = FQ_MUL(&R, &d), yh = FQ_MUL(&p->Y, &HHH);
  r.Y = FQ_SUB(&rd, &yh);
  r.Z = FQ_MUL(&p->Z, &H);
  return r;
}
/* a^-1 mod m by the binary extended Euclid (HAC 14.61 for an odd modulus), as a bignum library's modular inverse
 * would (galois-field's `recip` on a prime field goes through GMP, not through a^(m-2)); 0 for a = 0 */
static void shr1(fe* a, uint64_t top) {
  a->v[0] = (a->v[0] >> 1) | (a->v[1] << 63);
  a->v[1] = (a->v[1] >> 1) | (a->v[2] << 63);
  a->v[2] = (a->v[2] >> 1) | (a->v[3] << 63);
  a->v[3] = (a->v[3] >> 1) | (top << 63);
}
static void halve_mod(fe* x, const fe* m) {
  uint64_t top = 0;
  if (x->v[0] & 1) { fe t; top = fe_add_raw(&t, x, m); *x = t; }
  shr1(x, top);
}
static fe mod_inv_euclid(const fe* a, const fe* m) {
  fe u = *a, v = *m, x1 = fe_from_u64(1), x2 = fe_from_u64(0), one = fe_from_u64(1), t;
  if (fe_is_zero(a)) return x2;
  while (fe_cmp(&u, &one) != 0 && fe_cmp(&v, &one) != 0) {
    while (!(u.v[0] & 1)) { shr1(&u, 0); halve_mod(&x1, m); }
    while (!(v.v[0] & 1)) { shr1(&v, 0); halve_mod(&x2, m); }
    if (fe_cmp(&u, &v) >= 0) { fe_sub_raw(&t, &u, &v); u = t; x1 = mod_sub(&x1, &x2, m); }
    else { fe_sub_raw(&t, &v, &u); v = t; x2 = mod_sub(&x2, &x1, m); }
  }
  return fe_cmp(&u, &one) == 0 ? x1 : x2;
}
/* `<>` on affine points with the library-style inverse: the step of the literal affine ladder below */
static pt pt_add_affine_lit(const pt* a, const pt* b) {
  if (a->inf) return *b;
  if (b->inf) return *a;
  fe lam;
  if (fe_cmp(&a->x, &b->x) == 0) {
    fe s = FQ_ADD(&a->y, &b->y);
    if (fe_is_zero(&s)) return pt_inf();
    fe xx = FQ_MUL(&a->x, &a->x), three = fe_from_u64(3), num = FQ_MUL(&xx, &three);
    fe den = FQ_ADD(&a->y, &a->y), di = mod_inv_euclid(&den, &FP);
    lam = FQ_MUL(&num, &di);
  } else {
    fe num = FQ_SUB(&b->y, &a->y), den = FQ_SUB(&b->x, &a->x), di = mod_inv_euclid(&den, &FP);
    lam = FQ_MUL(&num, &di);
  }
  pt r;
  r.inf = 0;
  fe l2 = FQ_MUL(&lam, &lam), t = FQ_SUB(&l2, &a->x);
  r.x = FQ_SUB(&t, &b->x);
  fe d = FQ_SUB(&a->x, &r.x), ld = FQ_MUL(&lam, &d);
  r.y = FQ_SUB(&ld, &a->y);
  return r;
}
/* 0: products in Jacobian coordinates with one inversion at the end (generous to the reference);
 * 1: the LITERAL affine double-and-add that `mul` on a `PA` performs (Data.Curve.Weierstrass affine `dbl` / `add`:
 *    one field inversion per doubling and per addition).  Same group elements either way. */
static int g_mul_mode = 0;
void bpref_set_mul_mode(int mode) { g_mul_mode = mode ? 1 : 0; }

/* `mul`: k * P by double-and-add, k canonical (< r) */
static pt pt_mul(const pt* p, const fe* k) {
  if (p->inf || fe_is_zero(k)) return pt_inf();
  if (g_mul_mode == 1) {
    pt acc = pt_inf();
    for (int i = 255; i >= 0; i--) {
      if (!acc.inf) acc = pt_add_affine_lit(&acc, &acc);
      if ((k->v[i >> 6] >> (i & 63)) & 1) acc = pt_add_affine_lit(&acc, p);
    }
    return acc;
  }
  jac acc;
  memset(&acc, 0, sizeof acc);
  acc.Y = fe_from_u64(1);
  int started = 0;
  for (int i = 255; i >= 0; i--) {
    if (started) acc = jac_dbl(&acc);
    if ((k->v[i >> 6] >> (i & 63)) & 1) { acc = jac_add_affine(&acc, p); started = 1; }
  }
  if (fe_is_zero(&acc.Z)) return pt_inf();
  fe zi = mod_inv(&acc.Z, &FP), zi2 = FQ_MUL(&zi, &zi), zi3 = FQ_MUL(&zi2, &zi);
  pt r;
  r.inf = 0;
  r.x = FQ_MUL(&acc.X, &zi2);
  r.y = FQ_MUL(&acc.Y, &zi3);
  return r;
}
/* `addTwoMulP` (Utils.hs:72-73) and `sumExps` (Utils.hs:76-80) */
static pt add_two_mul(const fe* e0, const pt* p0, const fe* e1, const pt* p1) {
  pt a = pt_mul(p0, e0), b = pt_mul(p1, e1);
  return pt_add(&a, &b);
}
static pt sum_exps(const fe* es, const pt* ps, size_t n) {
  pt acc = pt_inf();
  size_t i = n - (n % 2);
  if (n % 2) acc = pt_mul(&ps[n - 1], &es[n - 1]);
  while (i > 0) {
    i -= 2;
    pt t = add_two_mul(&es[i], &ps[i], &es[i + 1], &ps[i + 1]);
    acc = pt_add(&t, &acc);
  }
  return acc;
}

/* ---------------- SHA-256 + transcript ---------------- */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
    0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
    0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
    0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
    0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
    0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256(const uint8_t* msg, size_t len, uint8_t out[32]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t total = ((len + 9 + 63) / 64) * 64;
  uint8_t* buf = (uint8_t*)calloc(total, 1);
  memcpy(buf, msg, len);
  buf[len] = 0x80;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) buf[total - 1 - i] = (uint8_t)(bits >> (8 * i));
  for (size_t off = 0; off < total; off += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
      w[i] = ((uint32_t)buf[off + 4 * i] << 24) | ((uint32_t)buf[off + 4 * i + 1] << 16) |
             ((uint32_t)buf[off + 4 * i + 2] << 8) | buf[off + 4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t t1 = hh + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
      uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  free(buf);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(h[i] >> 24); out[4 * i + 1] = (uint8_t)(h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(h[i] >> 8); out[4 * i + 3] = (uint8_t)h[i];
  }
}

static int g_style = 0;        /* 0 derived Show, 1 plain */
static int g_sqrt_branch = 0;  /* 0: a^((p+1)/4), 1: the other root */

/* decimal text of a 256-bit integer (schoolbook division by 10) */
static size_t dec(const fe* a, char* out) {
  fe t = *a;
  char tmp[80];
  int n = 0;
  if (fe_is_zero(&t)) { out[0] = '0'; return 1; }
  while (!fe_is_zero(&t)) {
    u128 rem = 0;
    for (int i = 3; i >= 0; i--) { u128 cur = (rem << 64) | t.v[i]; t.v[i] = (uint64_t)(cur / 10); rem = cur % 10; }
    tmp[n++] = (char)('0' + (int)rem);
  }
  for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
  return (size_t)n;
}
static size_t show_fr(const fe* s, char* out) {
  size_t n = 0;
  if (g_style == 0) { out[n++] = 'P'; out[n++] = ' '; }
  return n + dec(s, out + n);
}
static size_t show_pt(const pt* p, char* out) {
  size_t n = 0;
  if (p->inf) { out[0] = 'O'; return 1; }
  if (g_style == 0) {
    memcpy(out, "A (P ", 5); n = 5; n += dec(&p->x, out + n);
    memcpy(out + n, ") (P ", 5); n += 5; n += dec(&p->y, out + n);
    out[n++] = ')';
  } else {
    out[n++] = 'A'; out[n++] = ' '; n += dec(&p->x, out + n); out[n++] = ' '; n += dec(&p->y, out + n);
  }
  return n;
}
static const char* SHOW_R = "115792089237316195423570985008687907852837564279074904382605163141518161494337";
/* `oracle` (Utils.hs:29-30) */
static fe oracle_mod(const char* msg, size_t len, const fe* m) {
  uint8_t d[32];
  sha256((const uint8_t*)msg, len, d);
  fe v;
  for (int i = 0; i < 4; i++) {
    uint64_t w = 0;
    for (int j = 0; j < 8; j++) w = (w << 8) | d[8 * (3 - i) + j];
    v.v[i] = w;
  }
  fe s;
  while (fe_cmp(&v, m) >= 0) { fe_sub_raw(&s, &v, m); v = s; }
  return v;
}
static fe shamir_y(const pt* a, const pt* s) {
  char b[1024]; size_t n = 0;
  n += (size_t)sprintf(b, "%s", SHOW_R); n += show_pt(a, b + n); n += show_pt(s, b + n);
  return oracle_mod(b, n, &FR);
}
static fe shamir_z(const pt* a, const pt* s, const fe* y) {
  char b[1024]; size_t n = 0;
  n += (size_t)sprintf(b, "%s", SHOW_R); n += show_pt(a, b + n); n += show_pt(s, b + n); n += show_fr(y, b + n);
  return oracle_mod(b, n, &FR);
}
static fe shamir_x(const pt* a, const pt* s, const pt* t1, const pt* t2, const fe* y, const fe* z) {
  char b[2048]; size_t n = 0;
  n += (size_t)sprintf(b, "%s", SHOW_R); n += show_pt(a, b + n); n += show_pt(s, b + n); n += show_pt(t1, b + n);
  n += show_pt(t2, b + n); n += show_fr(y, b + n); n += show_fr(z, b + n);
  return oracle_mod(b, n, &FR);
}
static fe shamir_x_prime(const pt* c, const pt* l, const pt* r) {
  char b[1024]; size_t n = 0;
  n += (size_t)sprintf(b, "%s", SHOW_R); n += show_pt(l, b + n); n += show_pt(r, b + n); n += show_pt(c, b + n);
  return oracle_mod(b, n, &FR);
}
static fe shamir_u(const fe* tb, const fe* mu, const fe* t) {
  char b[1024]; size_t n = 0;
  n += (size_t)sprintf(b, "%s", SHOW_R); n += show_fr(tb, b + n); n += show_fr(mu, b + n); n += show_fr(t, b + n);
  return oracle_mod(b, n, &FR);
}

/* ---------------- generators (Utils.hs:14-25,38-47) ---------------- */
static pt g_h;
static pt* g_gs = NULL;
static pt* g_hs = NULL;
static size_t g_ngens = 0;

static pt generate_h(void) {
  char extra[300] = "";
  pt g = pt_gen();
  for (;;) {
    char b[1024]; size_t n = show_pt(&g, b);
    size_t el = strlen(extra);
    memcpy(b + n, extra, el); n += el;
    fe x = oracle_mod(b, n, &FP);
    fe xx = FQ_MUL(&x, &x), x3 = FQ_MUL(&xx, &x), seven = fe_from_u64(7), rhs = FQ_ADD(&x3, &seven);
    /* sr: p = 3 mod 4 */
    fe e = {{0xFFFFFFFFBFFFFF0CULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x3FFFFFFFFFFFFFFFULL}}; /* (p+1)/4 */
    fe y = mod_pow(&rhs, &e, &FP), yy = FQ_MUL(&y, &y);
    if (fe_cmp(&yy, &rhs) == 0) {
      pt r; r.inf = 0; r.x = x; r.y = g_sqrt_branch ? mod_neg(&y, &FP) : y;
      return r;
    }
    memmove(extra + 1, extra, el + 1);
    extra[0] = '1';
  }
}

static fe ld_fe(const uint8_t* b) { fe r; memcpy(r.v, b, 32); return r; }
static void st_fe(uint8_t* b, const fe* a) { memcpy(b, a->v, 32); }
static pt ld_pt(const uint8_t* b) {
  pt p; p.x = ld_fe(b); p.y = ld_fe(b + 32);
  p.inf = fe_is_zero(&p.x) && fe_is_zero(&p.y);
  return p;
}
static void st_pt(uint8_t* b, const pt* p) {
  if (p->inf) { memset(b, 0, 64); return; }
  st_fe(b, &p->x); st_fe(b + 32, &p->y);
}

void bpref_init(int style, int sqrt_branch, uint32_t n_gens) {
  g_style = style;
  g_sqrt_branch = sqrt_branch;
  free(g_gs); free(g_hs);
  g_gs = (pt*)malloc(sizeof(pt) * n_gens);
  g_hs = (pt*)malloc(sizeof(pt) * n_gens);
  g_ngens = n_gens;
  g_h = generate_h();
  pt g = pt_gen();
  for (uint32_t i = 0; i < n_gens; i++) {
    char b[1024];
    size_t n = (size_t)sprintf(b, "%u", i + 1);
    size_t n1 = n + show_pt(&g, b + n);
    fe k = oracle_mod(b, n1, &FR);
    g_gs[i] = pt_mul(&g, &k);
    size_t n2 = n + show_pt(&g_h, b + n);
    k = oracle_mod(b, n2, &FR);
    g_hs[i] = pt_mul(&g, &k);
  }
}
void bpref_get_generators(uint8_t* gs, uint8_t* hs, uint8_t* h) {
  for (size_t i = 0; i < g_ngens; i++) { st_pt(gs + 64 * i, &g_gs[i]); st_pt(hs + 64 * i, &g_hs[i]); }
  st_pt(h, &g_h);
}
void bpref_point_mul(const uint8_t* p, const uint8_t* k, uint8_t* out) {
  pt a = ld_pt(p); fe s = ld_fe(k);
  fe t; while (fe_cmp(&s, &FR) >= 0) { fe_sub_raw(&t, &s, &FR); s = t; }
  pt r = pt_mul(&a, &s);
  st_pt(out, &r);
}
/* sumExps over n (scalar, point) pairs in the engine's byte encoding */
void bpref_sum_exps(const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t* out) {
  fe* es = (fe*)malloc(sizeof(fe) * (n ? n : 1));
  pt* ps = (pt*)malloc(sizeof(pt) * (n ? n : 1));
  for (size_t i = 0; i < n; i++) {
    es[i] = ld_fe(scalars + 32 * i);
    fe t; while (fe_cmp(&es[i], &FR) >= 0) { fe_sub_raw(&t, &es[i], &FR); es[i] = t; }
    ps[i] = ld_pt(points + 64 * i);
  }
  pt r = sum_exps(es, ps, n);
  st_pt(out, &r);
  free(es); free(ps);
}

/* `logBase2` (Utils.hs:93-94) */
static int log_base2_u(double x) { return (int)floor(log(x) / log(2.0)); }

/* powerVector (Utils.hs:50-52) */
static void power_vector(const fe* a, size_t n, fe* out) {
  fe cur = fe_from_u64(1);
  int a_zero = fe_is_zero(a);
  for (size_t i = 0; i < n; i++) {
    if (i == 0 && a_zero) out[i] = fe_from_u64(0); else out[i] = cur;
    cur = mod_mul(&cur, a, &FR);
  }
}

/* MultiRangeProof.verifyProof for proof `p` of a batch in the engine's SoA layout
 * (scalars: 5 rows [B][32]: tBlinding, mu, t, l, r; points: A, S, T1, T2, L[k], R[k], V[m]). */
static int verify_one(uint32_t n, uint32_t m, size_t B, const uint8_t* sc, const uint8_t* pts, size_t p) {
  uint32_t nm = n * m, k = 0;
  while ((1u << k) < nm) k++;
#define SC(r) ld_fe(sc + ((size_t)(r) * B + p) * 32)
#define PT(r) ld_pt(pts + ((size_t)(r) * B + p) * 64)
  fe tb = SC(0), mu = SC(1), t = SC(2), l = SC(3), r = SC(4);
  pt A = PT(0), S = PT(1), T1 = PT(2), T2 = PT(3);
  fe y = shamir_y(&A, &S), z = shamir_z(&A, &S, &y), x = shamir_x(&A, &S, &T1, &T2, &y, &z);
  pt G = pt_gen();
  fe* yn = (fe*)malloc(sizeof(fe) * nm);
  fe* two = (fe*)malloc(sizeof(fe) * n);
  fe f2 = fe_from_u64(2), f1 = fe_from_u64(1);
  power_vector(&y, nm, yn);
  power_vector(&f2, n, two);
  fe z2 = mod_mul(&z, &z, &FR), x2 = mod_mul(&x, &x, &FR);
  /* verifyTPoly (Verifier.hs:44-65) */
  pt lhs = add_two_mul(&t, &G, &tb, &g_h);
  fe* zs = (fe*)malloc(sizeof(fe) * m);
  pt* vs = (pt*)malloc(sizeof(pt) * m);
  {
    fe* zp = (fe*)malloc(sizeof(fe) * m);
    power_vector(&z, m, zp);
    for (uint32_t j = 0; j < m; j++) { zs[j] = mod_mul(&z2, &zp[j], &FR); vs[j] = PT(4 + 2 * k + j); }
    free(zp);
  }
  pt rhs = sum_exps(zs, vs, m);
  /* delta (RangeProof/Internal.hs:152-157) */
  fe sy = fe_from_u64(0), s2 = fe_from_u64(0);
  for (uint32_t i = 0; i < nm; i++) sy = mod_add(&sy, &yn[i], &FR);
  for (uint32_t i = 0; i < n; i++) s2 = mod_add(&s2, &two[i], &FR);
  fe zz = mod_sub(&z, &z2, &FR), d1 = mod_mul(&zz, &sy, &FR), d2 = fe_from_u64(0), zj = mod_mul(&z2, &z, &FR);
  for (uint32_t j = 1; j <= m; j++) { fe tt = mod_mul(&zj, &s2, &FR); d2 = mod_add(&d2, &tt, &FR); zj = mod_mul(&zj, &z, &FR); }
  fe delta = mod_sub(&d1, &d2, &FR);
  pt gd = pt_mul(&G, &delta), t1x = pt_mul(&T1, &x), t2x = pt_mul(&T2, &x2);
  rhs = pt_add(&rhs, &gd); rhs = pt_add(&rhs, &t1x); rhs = pt_add(&rhs, &t2x);
  int ok_t = pt_eq(&lhs, &rhs);
  /* verifyLRCommitment (Verifier.hs:68-88): hs', u, computeLRCommitment */
  pt* hsp = (pt*)malloc(sizeof(pt) * nm);
  for (uint32_t i = 0; i < nm; i++) { fe yi = mod_inv(&yn[i], &FR); hsp[i] = pt_mul(&g_hs[i], &yi); }
  fe uch = shamir_u(&tb, &mu, &t);
  pt u = pt_mul(&G, &uch);
  pt gsum = pt_inf();
  for (uint32_t i = 0; i < nm; i++) gsum = pt_add(&gsum, &g_gs[i]);
  pt P = A;
  pt xs = pt_mul(&S, &x); P = pt_add(&P, &xs);
  pt zg = pt_mul(&gsum, &z); zg = pt_neg(&zg); P = pt_add(&P, &zg);
  fe* hexp = (fe*)malloc(sizeof(fe) * nm);
  for (uint32_t i = 0; i < nm; i++) hexp[i] = mod_mul(&z, &yn[i], &FR);
  pt he = sum_exps(hexp, hsp, nm); P = pt_add(&P, &he);
  pt inner = pt_inf();
  fe zj1 = z2; /* z^(j+1), j = 1 */
  for (uint32_t j = 1; j <= m; j++) {
    for (uint32_t i = 0; i < n; i++) hexp[i] = mod_mul(&zj1, &two[i], &FR);
    pt sj = sum_exps(hexp, hsp + (size_t)(j - 1) * n, n);
    inner = pt_add(&inner, &sj);
    zj1 = mod_mul(&zj1, &z, &FR);
  }
  P = pt_add(&P, &inner);
  pt mh = pt_mul(&g_h, &mu); mh = pt_neg(&mh); P = pt_add(&P, &mh);
  pt ut = pt_mul(&u, &t); P = pt_add(&P, &ut);
  /* IPP verifyProof (InnerProductProof/Verifier.hs:18-73) */
  fe* ch = (fe*)malloc(sizeof(fe) * (k ? k : 1)); /* challenges, last round first */
  pt acc = P;
  for (uint32_t rd = 0; rd < k; rd++) {
    pt L = PT(4 + rd), Rr = PT(4 + k + rd);
    fe xk = shamir_x_prime(&acc, &L, &Rr), xi = mod_inv(&xk, &FR);
    fe xk2 = mod_mul(&xk, &xk, &FR), xi2 = mod_mul(&xi, &xi, &FR);
    pt a1 = pt_mul(&L, &xk2), a2 = pt_mul(&Rr, &xi2);
    pt s = pt_add(&a1, &a2); acc = pt_add(&s, &acc);
    ch[k - 1 - rd] = xk;
  }
  /* mkOtherExponents: s_0 = 1/prod, s_{i + 2^j} = s_i * ch[j]^2, first insertion wins */
  fe* sv = (fe*)malloc(sizeof(fe) * nm);
  char* have = (char*)calloc(nm, 1);
  fe prod = f1;
  for (uint32_t j = 0; j < k; j++) prod = mod_mul(&prod, &ch[j], &FR);
  sv[0] = mod_inv(&prod, &FR); have[0] = 1;
  int lg = nm > 0 ? log_base2_u((double)nm) : 0;
  for (uint32_t i = 0; i < nm / 2; i++)
    for (int j = 0; j < lg; j++) {
      uint32_t i1 = (1u << j) + i;
      if (i1 >= nm || have[i1]) continue;
      fe c2 = mod_mul(&ch[j], &ch[j], &FR);
      sv[i1] = mod_mul(&sv[i], &c2, &FR); have[i1] = 1;
    }
  fe* srev = (fe*)malloc(sizeof(fe) * nm);
  for (uint32_t i = 0; i < nm; i++) srev[i] = sv[nm - 1 - i];
  pt gc = sum_exps(sv, g_gs, nm), hc = sum_exps(srev, hsp, nm);
  fe lr = mod_mul(&l, &r, &FR);
  pt c1 = pt_mul(&gc, &l), c2p = pt_mul(&hc, &r), c3 = pt_mul(&u, &lr);
  pt cp = pt_add(&c1, &c2p); cp = pt_add(&cp, &c3);
  int ok_lr = pt_eq(&acc, &cp);
  free(yn); free(two); free(zs); free(vs); free(hsp); free(hexp); free(ch); free(sv); free(have); free(srev);
  (void)f1;
  return ok_t && ok_lr;
#undef SC
#undef PT
}

/* verdicts[i] for proofs first..first+count-1 of a batch of B */
void bpref_verify_batch(uint32_t n_bits, uint32_t m, size_t B, const uint8_t* scalars, const uint8_t* points,
                        size_t first, size_t count, uint8_t* verdicts) {
  for (size_t i = 0; i < count; i++) verdicts[i] = (uint8_t)verify_one(n_bits, m, B, scalars, points, first + i);
}
