// Host build of the device arithmetic headers (fe.h, ec.h) for CPU-side tests.
// TEST INFRASTRUCTURE: lets `pytest -m "not gpu"` check the exact source the
// HIP kernels compile against the oracle.  Not part of the product path.
#include <string.h>
#include "../../bulletproofs_amd/csrc/ec.h"
using namespace bp;

static u256 ld(const uint8_t* p) { u256 r; memcpy(r.v, p, 32); return r; }
static void st(uint8_t* p, const u256& a) { memcpy(p, a.v, 32); }
static affine lda(const uint8_t* p) { affine a; a.x = ld(p); a.y = ld(p + 32); return a; }
static void sta(uint8_t* p, const affine& a) { st(p, a.x); st(p + 32, a.y); }

extern "C" {
// op: 0 add 1 sub 2 mul 3 sqr 4 inv 5 sqrt-candidate 6 neg; inputs may be weakly reduced
void hs_fq_op(int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  u256 x = ld(a), y = ld(b), r;
  switch (op) {
    case 0: r = fq_add(x, y); break;
    case 1: r = fq_sub(x, y); break;
    case 2: r = fq_mul(x, y); break;
    case 3: r = fq_sqr(x); break;
    case 4: r = fq_inv(x); break;
    case 5: r = fq_sqrt_candidate(x); break;
    default: r = fq_neg(x); break;
  }
  st(out, fq_canon(r));
}
int hs_fq_is_zero(const uint8_t* a) { return fq_is_zero(ld(a)); }
// generic Montgomery arm; which: 0 = Fr, 1 = Fq.  canonical in/out (not Montgomery form)
void hs_mont_op(int which, int op, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  u256 x = ld(a), y = ld(b), r;
  if (which == 0) {
    x = Fr::to_mont(x); y = Fr::to_mont(y);
    switch (op) {
      case 0: r = Fr::add(x, y); break;
      case 1: r = Fr::sub(x, y); break;
      case 2: r = Fr::mul(x, y); break;
      case 3: r = Fr::sqr(x); break;
      case 4: r = Fr::inv(x); break;
      default: r = Fr::neg(x); break;
    }
    st(out, Fr::from_mont(r));
  } else {
    x = FqMont::to_mont(x); y = FqMont::to_mont(y);
    switch (op) {
      case 0: r = FqMont::add(x, y); break;
      case 1: r = FqMont::sub(x, y); break;
      case 2: r = FqMont::mul(x, y); break;
      case 3: r = FqMont::sqr(x); break;
      case 4: r = FqMont::inv(x); break;
      default: r = FqMont::neg(x); break;
    }
    st(out, FqMont::from_mont(r));
  }
}
// point ops on affine 64-byte LE (x||y), identity = all zero
// op: 0 add (via xyzz_add) 1 madd 2 dbl 3 neg 4 mul by scalar k
void hs_point_op(int op, const uint8_t* p, const uint8_t* q, const uint8_t* k, uint8_t* out) {
  affine a = lda(p), b = lda(q);
  xyzz r;
  switch (op) {
    case 0: r = xyzz_add(xyzz_from_affine(a), xyzz_from_affine(b)); break;
    case 1: r = xyzz_madd(xyzz_from_affine(a), b); break;
    case 2: r = xyzz_dbl(xyzz_from_affine(a)); break;
    case 3: r = xyzz_from_affine(affine_neg(a)); break;
    default: r = xyzz_mul_affine(a, ld(k)); break;
  }
  sta(out, xyzz_to_affine(r));
}
// (k1*P) + (k2*Q) with both partial results left un-normalised (exercises add on general XYZZ)
void hs_point_lincomb(const uint8_t* p, const uint8_t* k1, const uint8_t* q, const uint8_t* k2, uint8_t* out) {
  xyzz a = xyzz_mul_affine(lda(p), ld(k1));
  xyzz b = xyzz_mul_affine(lda(q), ld(k2));
  sta(out, xyzz_to_affine(xyzz_add(a, b)));
}
int hs_on_curve(const uint8_t* p) { return affine_on_curve(lda(p)); }
}

// ---- serial simulation of the bucket MSM built from msm_core.h bodies ---------
#include <vector>
#include "../../bulletproofs_amd/csrc/msm_core.h"
extern "C" void hs_msm(const uint8_t* scalars, const uint8_t* points, size_t n, int c, uint8_t* out) {
  MsmPlan pl = msm_plan(c);
  std::vector<xyzz> buckets((size_t)pl.W * pl.nb, xyzz_inf());
  for (size_t i = 0; i < n; i++) {
    u256 s = ld(scalars + 32 * i);
    affine p = lda(points + 64 * i);
    uint32_t carry = 0;
    for (int w = 0; w < pl.W; w++) {
      uint32_t code = recode_next(s, c, carry);
      if (code == MSM_SKIP) continue;
      bucket_add(buckets[(size_t)w * pl.nb + (code & ~MSM_NEG)], p, (code & MSM_NEG) != 0);
    }
  }
  std::vector<xyzz> wsum(pl.W);
  for (int w = 0; w < pl.W; w++) {
    xyzz acc = xyzz_inf();
    for (int sg = 0; sg < pl.nseg; sg++) {
      xyzz S, T;
      const xyzz* base = &buckets[(size_t)w * pl.nb + (size_t)sg * pl.seg];
      segment_reduce(pl.seg, [&](int j) { return base[j]; }, S, T);
      acc = xyzz_add(acc, segment_value(pl, sg, S, T));
    }
    wsum[w] = acc;
  }
  xyzz r = window_combine(pl, [&](int w) { return wsum[w]; });
  sta(out, xyzz_to_affine(r));
}

// ---- GLV decomposition + joint-table multi-exponentiation (glv.h) ----------------------------
#include "../../bulletproofs_amd/csrc/glv.h"
// out1/out2: 32-byte magnitudes, signs in sgn[0], sgn[1]
extern "C" void hs_glv_split(const uint8_t* k, uint8_t* out1, uint8_t* out2, uint8_t* sgn) {
  GlvHalf h1, h2;
  glv_split(ld(k), h1, h2);
  memset(out1, 0, 32); memset(out2, 0, 32);
  memcpy(out1, h1.m, 20); memcpy(out2, h2.m, 20);
  sgn[0] = h1.neg; sgn[1] = h2.neg;
}
// np = 1: k0*P0 ; np = 2: k0*P0 + k1*P1
extern "C" void hs_glv_multiexp(int np, const uint8_t* ks, const uint8_t* ps, uint8_t* out) {
  xyzz tab[16];
  xyzz r;
  if (np == 1) {
    u256 k[1] = {ld(ks)}; affine P[1] = {lda(ps)};
    r = glv_multiexp_local<1>(k, P, tab);
  } else {
    u256 k[2] = {ld(ks), ld(ks + 32)}; affine P[2] = {lda(ps), lda(ps + 64)};
    r = glv_multiexp_local<2>(k, P, tab);
  }
  sta(out, xyzz_to_affine(r));
}

// non-negative halves (32-byte little-endian each)
extern "C" void hs_glv_split_pos(const uint8_t* k, uint8_t* out1, uint8_t* out2) {
  GlvPos g = glv_split_pos(ld(k));
  memset(out1, 0, 32); memset(out2, 0, 32);
  memcpy(out1, g.m1, 20); memcpy(out2, g.m2, 20);
}
// k0*P0 + k1*P1 through the sign-free affine joint table of the pair
extern "C" void hs_glv_pair_tab(const uint8_t* ks, const uint8_t* ps, uint8_t* out) {
  affine tab[16];
  glv_pair_table_local(lda(ps), lda(ps + 64), tab);
  xyzz r = glv_pair_multiexp_tab(ld(ks), ld(ks + 32), [&](int e) { return tab[e]; });
  sta(out, xyzz_to_affine(r));
}

// k*P as the verifier's latency path forms it: the bit table 2^j P, one glv_bit_lane term per j, summed
extern "C" void hs_glv_bit_lanes(const uint8_t* k, const uint8_t* pt, uint8_t* out) {
  GlvHalf h1, h2;
  glv_split(ld(k), h1, h2);
  xyzz t = xyzz_from_affine(lda(pt)), acc = xyzz_inf();
  for (int j = 0; j < GLV_LANE_BITS; j++) {
    acc = xyzz_add(acc, glv_bit_lane(h1, h2, t, j));
    t = xyzz_dbl(t);
  }
  sta(out, xyzz_to_affine(acc));
}

// ---- 29-bit-limb base field and the lazy mixed addition of the accumulation loops (fq29.h) ----
#include "../../bulletproofs_amd/csrc/fq29.h"
// op 0: a*b  1: a^2  2: a*b + c*d  3: round trip; operands canonical, result canonical
extern "C" void hs_fe29_op(int op, const uint8_t* a, const uint8_t* b, const uint8_t* c, const uint8_t* d, uint8_t* out) {
  fe29 x = fe29_from_u256(ld(a)), y = fe29_from_u256(ld(b)), r;
  switch (op) {
    case 0: r = fe29_mul(x, y); break;
    case 1: r = fe29_sqr(x); break;
    case 2: r = fe29_mul2(x, y, fe29_from_u256(ld(c)), fe29_from_u256(ld(d))); break;
    case 4: st(out, fq_canon(fq_inv29(ld(a)))); return;
    default: r = x; break;
  }
  st(out, fq_canon(fe29_to_u256(r)));
}
// lazy operands at the documented bounds: (a + (64p - c)) * norm(b + (64p - c)) + (64p - a) * b, and
// norm(a + (64p - b))^2 + (128p - (b + 2c))  -> canonical
extern "C" void hs_fe29_lazy(const uint8_t* a, const uint8_t* b, const uint8_t* c, uint8_t* out, uint8_t* out2) {
  fe29 x = fe29_from_u256(ld(a)), y = fe29_from_u256(ld(b)), z = fe29_from_u256(ld(c));
  fe29 l = fe29_add(x, fe29_neg(z));
  fe29 r = fe29_norm(fe29_add(y, fe29_neg(z)));
  st(out, fq_canon(fe29_to_u256(fe29_mul2(r, l, fe29_neg(x), y))));
  fe29 w;
  for (int i = 0; i < 9; i++) w.v[i] = y.v[i] + (z.v[i] << 1);
  st(out2, fq_canon(fe29_to_u256(fe29_sqr_add(fe29_norm(fe29_add(x, fe29_neg(y))), fe29_neg2(w)))));
}
// sum of n points (+- by neg[i]) accumulated from the identity with xyzz29_madd
extern "C" void hs_madd29_chain(const uint8_t* pts, const uint8_t* neg, size_t n, uint8_t* out) {
  xyzz29 acc = xyzz29_from(xyzz_inf());
  for (size_t i = 0; i < n; i++) acc = xyzz29_madd(acc, lda(pts + i * 64), neg[i] != 0);
  sta(out, xyzz_to_affine(xyzz29_to(acc)));
}

// ---- host-side inversion of the latency paths (host_inv.h) ----
#include "../../bulletproofs_amd/csrc/host_inv.h"
// which: 0 = 1/a mod p (weak input allowed), 1 = x^-1 mod r through the Montgomery-form wrapper, 2 = host Fr4 fr_inv
extern "C" void hs_host_inv(int which, const uint8_t* a, uint8_t* out) {
  if (which == 0) {
    st(out, bp::host::fq_inv_host(ld(a)));
  } else if (which == 1) {
    st(out, Fr::from_mont(bp::host::fr_inv_mont_host(Fr::to_mont(ld(a)))));
  } else {
    bp::host::fr_to_bytes(bp::host::fr_inv(bp::host::fr_from_bytes(a)), out);
  }
}
extern "C" void hs_xyzz_to_affine_host(const uint8_t* xyzz128, uint8_t* out) {
  xyzz p;
  p.X = ld(xyzz128); p.Y = ld(xyzz128 + 32); p.ZZ = ld(xyzz128 + 64); p.ZZZ = ld(xyzz128 + 96);
  sta(out, bp::host::xyzz_to_affine_host(p));
}

// the same through the 29-bit-limb chain (glv_pair_multiexp_tab29)
extern "C" void hs_glv_pair_tab29(const uint8_t* ks, const uint8_t* ps, uint8_t* out) {
  affine tab[16];
  glv_pair_table_local(lda(ps), lda(ps + 64), tab);
  xyzz r = glv_pair_multiexp_tab29(ld(ks), ld(ks + 32), [&](int e) { return tab[e]; });
  sta(out, xyzz_to_affine(r));
}
// P + phi(P) by the closed form of glv.h
extern "C" void hs_glv_p_plus_phi(const uint8_t* pt, uint8_t* out) {
  affine r = glv_p_plus_phi(lda(pt));
  r.x = fq_canon(r.x);
  r.y = fq_canon(r.y);
  sta(out, r);
}
// two joint tables on one doubling chain (glv_tables_multiexp29): ks = 4 scalars, present = 4 flags, ps = 4 points; an
// absent term's point is not read (its table half is left as the identity)
extern "C" void hs_glv_tables29(int nt, const uint8_t* ks, const uint8_t* present, const uint8_t* ps, uint8_t* out) {
  affine tab[2][16];
  for (int t = 0; t < nt; t++)
    glv_pair_table_local(lda(ps + (2 * t) * 64), present[2 * t + 1] ? lda(ps + (2 * t + 1) * 64) : affine_inf(), tab[t]);
  auto load = [&](int t, int e) { return tab[t][e]; };
  xyzz r;
  if (nt == 1) {
    const u256 k[2] = {ld(ks), ld(ks + 32)};
    const bool pr[2] = {present[0] != 0, present[1] != 0};
    r = glv_tables_multiexp29<1>(k, pr, load);
  } else {
    const u256 k[4] = {ld(ks), ld(ks + 32), ld(ks + 64), ld(ks + 96)};
    const bool pr[4] = {present[0] != 0, present[1] != 0, present[2] != 0, present[3] != 0};
    r = glv_tables_multiexp29<2>(k, pr, load);
  }
  sta(out, xyzz_to_affine(r));
}
// sum of n points by full XYZZ additions on lazy limbs (every point enters as k_i * P_i with small k_i so that Z != 1)
extern "C" void hs_add29_chain(const uint8_t* pts, size_t n, uint8_t* out) {
  xyzz29 acc = xyzz29_inf();
  for (size_t i = 0; i < n; i++) {
    xyzz29 q = xyzz29_dbl(xyzz29_dbl(xyzz29_from_affine(lda(pts + i * 64), false)));  // 4 P_i, projective
    acc = xyzz29_add(acc, q);
  }
  sta(out, xyzz_to_affine(xyzz29_to(acc)));
}
// n doublings of a point on lazy limbs
extern "C" void hs_dbl29_chain(const uint8_t* pt, int n, uint8_t* out) {
  xyzz29 acc = xyzz29_from_affine(lda(pt), false);
  for (int i = 0; i < n; i++) acc = xyzz29_dbl(acc);
  sta(out, xyzz_to_affine(xyzz29_to(acc)));
}

// ---- safegcd inversion (modinv30.h) ----
#include "../../bulletproofs_amd/csrc/modinv30.h"
// which: 0 base field (weakly reduced input), 1 scalar field (canonical, plain), 2 scalar field in Montgomery form
// (out = fr_inv_gcd(to_mont(a)) taken back out of Montgomery form)
extern "C" void hs_modinv30(int which, const uint8_t* a, uint8_t* out) {
  if (which == 0) st(out, fq_canon(fq_inv_gcd(ld(a))));
  if (which == 1) st(out, modinv30(ld(a), modinv30_r()));
  if (which == 2) st(out, Fr::from_mont(fr_inv_gcd(Fr::to_mont(ld(a)))));
}

// ---- device transcript (fsdev.h): the challenges of the range-proof verifier as the kernels form them ----
#include "../../bulletproofs_amd/csrc/fsdev.h"
// which: 0 shamirY (A,S)  1 shamirZ (A,S | y)  2 shamirX (A,S,T1,T2 | y,z)  3 shamirU (| tau,mu,t)  4 shamirX' (L,R,acc)
// pts: up to 4 points (64 B each), scs: up to 3 scalars; stride: words between consecutive words of the message buffer
extern "C" void hs_fs_challenge(int style, int which, const uint8_t* pts, const uint8_t* scs, size_t stride, uint8_t* out) {
  std::vector<uint32_t> buf((size_t)FS_MSG_WORDS * stride, 0xA5A5A5A5u);
  FsMsg m = fs_open(buf.data(), stride);
  auto pt = [&](int i) { fs_put_point(m, style, ld(pts + i * 64), ld(pts + i * 64 + 32)); };
  auto sc = [&](int i) { fs_put_fr(m, style, ld(scs + i * 32)); };
  if (which == 0) { pt(0); pt(1); }
  if (which == 1) { pt(0); pt(1); sc(0); }
  if (which == 2) { pt(0); pt(1); pt(2); pt(3); sc(0); sc(1); }
  if (which == 3) { sc(0); sc(1); sc(2); }
  if (which == 4) { pt(0); pt(1); pt(2); }
  st(out, fs_finish_fr(m));
}
// y, z, x as k_fs_initial forms them (shared opening hashed once), u, and a round's challenge in its two parts
static FsPoint ldfp(const uint8_t* p) { return FsPoint{ld(p), ld(p + 32)}; }
extern "C" void hs_fs_yzx(int style, const uint8_t* pts, size_t stride, uint8_t* out3) {
  std::vector<uint32_t> buf((size_t)FS_MSG_WORDS * stride, 0x5A5A5A5Au);
  u256 y, z, x;
  fs_challenges_yzx(style, ldfp(pts), ldfp(pts + 64), ldfp(pts + 128), ldfp(pts + 192), buf.data(), stride, y, z, x);
  st(out3, y);
  st(out3 + 32, z);
  st(out3 + 64, x);
}
extern "C" void hs_fs_u(int style, const uint8_t* scs, size_t stride, uint8_t* out) {
  std::vector<uint32_t> buf((size_t)FS_MSG_WORDS * stride, 0x5A5A5A5Au);
  st(out, fs_challenge_u(style, ld(scs), ld(scs + 32), ld(scs + 64), buf.data() + FS_U_WORD * stride, stride));
}
extern "C" void hs_fs_round(int style, const uint8_t* pts /*L, R, acc*/, size_t stride, size_t pstride, uint8_t* out) {
  std::vector<uint32_t> buf((size_t)FS_MSG_WORDS * stride, 0x5A5A5A5Au), buf2((size_t)FS_MSG_WORDS * stride, 0xC3C3C3C3u);
  std::vector<uint32_t> pre((size_t)FS_PRE_WORDS * pstride, 0x77777777u);
  fs_round_opening(style, ldfp(pts), ldfp(pts + 64), buf.data(), stride, pre.data(), pstride);
  st(out, fs_round_challenge(style, pre.data(), pstride, ldfp(pts + 128), buf2.data(), stride));
}

// ---- device decimal formatter (dec.h): 80-byte record of a 32-byte LE integer ----
#include "../../bulletproofs_amd/csrc/dec.h"
extern "C" void hs_dec_record(const uint8_t* le, uint8_t* rec80) {
  uint32_t w[DEC_REC_WORDS];
  u256_to_dec_record(ld(le), w);
  memcpy(rec80, w, DEC_REC);
}

// ---- host transcript primitives (transcript.h): paired SHA-256 and the decimal writer ----
#include "../../bulletproofs_amd/csrc/transcript.h"
// digests of two messages through the interleaved path (skip_a / skip_b full leading blocks are absorbed
// first, as the batch verifier does with shared prefixes); the buffers need 72 bytes of slack
extern "C" void hs_sha256_pair(uint8_t* a, size_t la, size_t skip_a, uint8_t* b, size_t lb, size_t skip_b, uint8_t* outa,
                               uint8_t* outb) {
  using namespace bp::host;
  ShaMid ma = sha_mid_init(), mb = sha_mid_init();
  sha_mid_advance(ma, a, skip_a * 64 < la ? skip_a * 64 : 0);
  sha_mid_advance(mb, b, skip_b * 64 < lb ? skip_b * 64 : 0);
  sha256_resume_inplace_x2(ma, a, la, outa, mb, b, lb, outb);
}
extern "C" size_t hs_u256_to_dec(const uint8_t* le, char* out) { return bp::host::u256_to_dec(le, out); }
// show text of a point / scalar from dec.h records (x record, y record), as the host appends it to a transcript
extern "C" size_t hs_show_point_rec(int style, const uint8_t* recs, char* out) {
  return bp::host::show_point_rec((bp::host::ShowStyle)style, recs, recs + 80, out);
}
extern "C" size_t hs_show_point(int style, const uint8_t* pt, char* out) {
  return bp::host::show_point((bp::host::ShowStyle)style, pt, out);
}

// ---- frvec.h: the per-element bodies of the scalar-field vector kernels (frvec.hip), run serially ---------------
#include "../../bulletproofs_amd/csrc/frvec.h"
// canonical bytes in / out; Montgomery form inside, as on the device
static u256 ldm(const uint8_t* p) { return fr_from_plain(ld(p)); }
static void stm(uint8_t* p, const u256& a) { st(p, Fr::from_mont(a)); }
extern "C" void hs_power_vector(const uint8_t* a, uint32_t n, uint8_t* out) {
  u256 am = ldm(a);
  for (uint32_t i = 0; i < n; i++) stm(out + 32 * (size_t)i, fr_power_elem(am, i));
}
extern "C" void hs_rp_polys(uint32_t n_bits, uint32_t nm, const uint8_t* bits, const uint8_t* s_r, const uint8_t* y,
                            const uint8_t* z, uint8_t* l0, uint8_t* r0, uint8_t* r1) {
  u256 ym = ldm(y), zm = ldm(z);
  for (uint32_t i = 0; i < nm; i++) {
    RpPolyElem e = rp_poly_elem(i, n_bits, bits[i], ldm(s_r + 32 * (size_t)i), ym, zm);
    stm(l0 + 32 * (size_t)i, e.l0);
    stm(r0 + 32 * (size_t)i, e.r0);
    stm(r1 + 32 * (size_t)i, e.r1);
  }
}
// one k_ipp_witness (fold part) + k_ipp_scalars pass over host arrays.  ls, rs: n canonical scalars (first nk_prev
// live), folded in place when fold != 0; cg, ch: n plain coefficients, updated in place; sc: [3][2n + 1][32] out
// (the bH entries are left untouched).
extern "C" void hs_ipp_round(uint32_t n, uint32_t nk_prev, int fold, const uint8_t* x, const uint8_t* xi, uint8_t* ls,
                             uint8_t* rs, uint8_t* cg, uint8_t* ch, uint8_t* sc) {
  std::vector<u256> l(n), r(n);
  for (uint32_t i = 0; i < nk_prev; i++) {
    l[i] = ldm(ls + 32 * (size_t)i);
    r[i] = ldm(rs + 32 * (size_t)i);
  }
  u256 xm = ldm(x), xim = ldm(xi);
  uint32_t nk = nk_prev;
  if (fold) {
    nk = nk_prev / 2;
    for (uint32_t j = 0; j < nk; j++) ipp_fold_elem(l[j], l[nk + j], r[j], r[nk + j], xm, xim);
  }
  const size_t nt = (size_t)2 * n + 1;
  for (uint32_t i = 0; i < n; i++) {
    u256 g = ld(cg + 32 * (size_t)i), h = ld(ch + 32 * (size_t)i);
    if (fold) ipp_coeff_elem(i, nk_prev, g, h, xm, xim);
    st(cg + 32 * (size_t)i, g);
    st(ch + 32 * (size_t)i, h);
    IppScalars s = ipp_scalar_elem(i, nk, g, h, [&](uint32_t j) { return l[j]; }, [&](uint32_t j) { return r[j]; });
    st(sc + 32 * (size_t)i, s.c_g);
    st(sc + 32 * ((size_t)n + i), s.c_h);
    st(sc + 32 * (nt + i), s.l_g);
    st(sc + 32 * (nt + n + i), s.l_h);
    st(sc + 32 * (2 * nt + i), s.r_g);
    st(sc + 32 * (2 * nt + n + i), s.r_h);
  }
  for (uint32_t i = 0; i < nk; i++) {
    stm(ls + 32 * (size_t)i, l[i]);
    stm(rs + 32 * (size_t)i, r[i]);
  }
}
// mkOtherExponents for n = 2^lg; challenges: k canonical scalars in ROUND order, k >= lg
extern "C" void hs_other_exponents(uint32_t lg, uint32_t k, const uint8_t* challenges, uint8_t* out) {
  u256 c2_rev[24], prod = Fr::one();
  for (uint32_t j = 0; j < k; j++) {
    u256 c = ldm(challenges + 32 * (size_t)(k - 1 - j));
    prod = Fr::mul(prod, c);
    if (j < 24) c2_rev[j] = Fr::sqr(c);
  }
  u256 s0 = Fr::inv(prod);
  for (uint32_t i = 0; i < (1u << lg); i++) stm(out + 32 * (size_t)i, ipp_other_exponent(i, lg, s0, c2_rev));
}
