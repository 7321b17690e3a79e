// Batched range-proof verification on gfx950: the EC and scalar-field work of
//   MultiRangeProof.verifyProof      (/root/reference/Bulletproofs/MultiRangeProof/Verifier.hs:19-88)
//   computeLRCommitment              (RangeProof/Internal.hs:169-206)
//   InnerProductProof.verifyProof    (InnerProductProof/Verifier.hs:18-73)
// for B independent proofs in lock-step.  Fiat-Shamir hashing stays on the host:
// the host supplies y, z, x, uChallenge and, round by round, x_k computed from the
// affine text of the running commitment this file returns (Verifier.hs:44-53).
//
// Per proof the verdict is  verifyTPoly && verifyLRCommitment  exactly: both are
// group equations, evaluated here as
//   E1 = sum_i (l s_i) G_i + sum_i (r s_{nm-1-i} y^-i) H_i + (l r u) G  -  acc_k   == O
//   E2 = (t - delta) G + tau H - sum_j z^(j+2) V_j - x T1 - x^2 T2               == O
// with acc_0 = P (computeLRCommitment) and acc_{k+1} = x_k^2 L_k + x_k^-2 R_k + acc_k.
// hs' = y^-i hs_i is never materialised: y^-i is folded into the scalars.
//
// Data layout: every per-proof array is [field][proof] (proof fastest) so a wave
// of 64 proofs reads and writes contiguous lines.
#include <string.h>
#include "engine.h"
#include "batchinv.h"
#include "dec.h"
#include "devio.h"
#include "fixedbase.h"
#include "fq29.h"
#include "lanes29.h"
#include "fsdev.h"
#include "glv.h"
#include "host_inv.h"
#include "verify.h"

namespace bp {

// ---- per-proof constants (Montgomery form), cst[c * B + p] -----------------------
enum { C_Y = 0, C_Z, C_X, C_U, C_YINV, C_X2, C_DELTA, C_S0, C_ZP /* z, plain form */, C_COUNT };

__device__ __forceinline__ u256 ldm(const u256* a, size_t idx) { return ld_u256(a + idx); }
__device__ __forceinline__ void stm(u256* a, size_t idx, const u256& v) { st_u256(a + idx, v); }

template <int WB>
__device__ __forceinline__ void put_digits_w(uint32_t* dig, uint32_t term, uint32_t B, uint32_t p, const u256& k_canon) {
  uint32_t d[fb_dwords(WB)];
  fb_recode<WB>(k_canon, d);
#pragma unroll
  for (int g = 0; g < fb_dwords(WB); g++) dig[((size_t)term * fb_dwords(WB) + g) * B + p] = d[g];
}
// digits of one term for the context's window width (uniform branch)
__device__ __forceinline__ void put_digits(uint32_t* dig, uint32_t term, const VDims& d, uint32_t p, const u256& k_canon) {
  if (d.wb == 20)
    put_digits_w<20>(dig, term, d.B, p, k_canon);
  else if (d.wb == 16)
    put_digits_w<16>(dig, term, d.B, p, k_canon);
  else if (d.wb == 12)
    put_digits_w<12>(dig, term, d.B, p, k_canon);
  else
    put_digits_w<8>(dig, term, d.B, p, k_canon);
}

// thread per proof: challenges -> Montgomery constants, y^-i ladder, z powers, delta
__global__ void __launch_bounds__(256) k_vprep0(VDims d, const uint8_t* __restrict__ chal /*[4][B][32]*/,
                                               u256* __restrict__ cst, u256* __restrict__ ypowinv /*[nm][B]*/,
                                               u256* __restrict__ zpow /*[m+3][B]*/, bool ladder) {
  __shared__ u256 inv_lds[BATCHINV_LDS];
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  size_t B = d.B;
  u256 y = Fr::one();
  if (p < d.B) y = Fr::to_mont(Fr::reduce_once(ld_u256(chal + ((size_t)0 * B + p) * 32)));
  u256 yinv = block_batch_inverse<FrInvOps>(y, inv_lds);  // one inversion per workgroup (batchinv.h)
  if (p >= d.B) return;
  u256 zp = Fr::reduce_once(ld_u256(chal + ((size_t)1 * B + p) * 32));
  u256 z = Fr::to_mont(zp);
  stm(cst, C_ZP * B + p, zp);
  u256 x = Fr::to_mont(Fr::reduce_once(ld_u256(chal + ((size_t)2 * B + p) * 32)));
  u256 u = Fr::to_mont(Fr::reduce_once(ld_u256(chal + ((size_t)3 * B + p) * 32)));
  stm(cst, C_Y * B + p, y);
  stm(cst, C_Z * B + p, z);
  stm(cst, C_X * B + p, x);
  stm(cst, C_U * B + p, u);
  stm(cst, C_YINV * B + p, yinv);
  stm(cst, C_X2 * B + p, Fr::sqr(x));
  // y^-i and sum_i y^i, i < nm   (powerVector y nm; RangeProof/Internal.hs:154)
  u256 ysum = u256_zero();
  if (ladder) {
    u256 cur = Fr::one(), ycur = Fr::one();
    for (uint32_t i = 0; i < d.nm; i++) {
      stm(ypowinv, (size_t)i * B + p, cur);
      cur = Fr::mul(cur, yinv);
      // powerVector's 0^0 = 0 special case (Utils.hs:50-52) only matters for y = 0
      ysum = Fr::add(ysum, ycur);
      ycur = Fr::mul(ycur, y);
    }
  } else {
    // few proofs, long vectors: k_vprep0_pow fills y^-i in parallel; the sum in closed form,
    // (y^nm - 1) / (y - 1), nm a power of two (the same residue as the term-by-term sum; y = 1 gives nm)
    u256 ynm = y;
    for (uint32_t j = 0; j < d.k; j++) ynm = Fr::sqr(ynm);
    u256 ym1 = Fr::sub(y, Fr::one());
    if (u256_is_zero(ym1)) {
      u256 nmv = u256_zero();
      nmv.v[0] = d.nm;
      ysum = Fr::to_mont(nmv);
    } else {
      ysum = Fr::mul(Fr::sub(ynm, Fr::one()), fr_inv_gcd(ym1));
    }
  }
  // z^e, e = 0..m+2
  u256 zc = Fr::one();
  for (uint32_t e = 0; e < d.m + 3; e++) {
    stm(zpow, (size_t)e * B + p, zc);
    zc = Fr::mul(zc, z);
  }
  // delta = (z - z^2) * sum y^i  -  sum_{j=1..m} z^(j+2) * (2^n - 1)
  u256 z2 = Fr::sqr(z);
  u256 two_n = u256_zero();
  two_n.v[d.n >> 5] = 1u << (d.n & 31);  // n <= 255
  u256 tn1 = Fr::sub(Fr::to_mont(Fr::reduce_once(two_n)), Fr::one());
  u256 zs = u256_zero(), zj = Fr::mul(z2, z);  // z^3
  for (uint32_t j = 0; j < d.m; j++) {
    zs = Fr::add(zs, zj);
    zj = Fr::mul(zj, z);
  }
  u256 delta = Fr::sub(Fr::mul(Fr::sub(z, z2), ysum), Fr::mul(zs, tn1));
  stm(cst, C_DELTA * B + p, delta);
}

// thread per (i, p): y^-i by square-and-multiply over the bits of i (small batches; see k_vprep0)
__global__ void __launch_bounds__(256) k_vprep0_pow(VDims d, const u256* __restrict__ cst, u256* __restrict__ ypowinv) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t B = d.B;
  if (t >= (size_t)d.nm * B) return;
  uint32_t i = (uint32_t)(t / B), p = (uint32_t)(t % B);
  u256 base = ldm(cst, C_YINV * B + p), r = Fr::one();
  for (uint32_t e = i; e; e >>= 1) {
    if (e & 1u) r = Fr::mul(r, base);
    base = Fr::sqr(base);
  }
  stm(ypowinv, (size_t)i * B + p, r);
}

// thread per (i, p): digits of c_i = z + z^(i/n + 2) * 2^(i%n) * y^-i for base H_i;
// i == 0 also emits the three single terms of P.
__global__ void __launch_bounds__(256) k_vprep1(VDims d, const uint8_t* __restrict__ sc /*[5][B][32]*/,
                                               const u256* __restrict__ cst, const u256* __restrict__ ypowinv,
                                               const u256* __restrict__ zpow, uint32_t* __restrict__ dig) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t B = d.B;
  if (t >= (size_t)d.nm * B) return;
  uint32_t i = (uint32_t)(t / B), p = (uint32_t)(t % B);
  u256 zp = ldm(cst, C_ZP * B + p);  // plain form
  uint32_t j = i / d.n, k = i % d.n;
  u256 two_k = u256_zero();
  two_k.v[k >> 5] = 1u << (k & 31);
  // a Montgomery product of a plain and a Montgomery-form operand is the plain product: two multiplications
  // give the canonical scalar directly (no to_mont / from_mont)
  u256 c = Fr::mul(Fr::mul(Fr::reduce_once(two_k), ldm(zpow, (size_t)(j + 2) * B + p)), ldm(ypowinv, (size_t)i * B + p));
  c = Fr::add(c, zp);
  put_digits(dig, i, d, p, c);
  if (i == 0) {
    u256 mu = Fr::reduce_once(ld_u256(sc + ((size_t)1 * B + p) * 32));
    u256 tt = Fr::to_mont(Fr::reduce_once(ld_u256(sc + ((size_t)2 * B + p) * 32)));
    put_digits(dig, d.nm + 0, d, p, Fr::neg(zp));                                       // -z  * GS
    put_digits(dig, d.nm + 1, d, p, Fr::neg(mu));                                       // -mu * H  (canonical neg)
    put_digits(dig, d.nm + 2, d, p, Fr::from_mont(Fr::mul(tt, ldm(cst, C_U * B + p))));  // t*u * G
  }
}

// ---- decimal text for the host's transcript (dec.h) ------------------------------------------------
// Every thread of the block holds NW dwords of records; the block's records are contiguous at `dst`.
// Staged through LDS so that consecutive lanes store consecutive 16-byte pieces: `dst` is pinned host
// memory, and lane-strided stores would cross the host link as 16-byte writes.
template <int NW>
__device__ __forceinline__ void block_store_records(uint32_t* lds, const uint32_t (&w)[NW], uint8_t* dst,
                                                    uint32_t valid_threads) {
#pragma unroll
  for (int i = 0; i < NW; i++) lds[threadIdx.x * NW + i] = w[i];
  __syncthreads();
  const uint4* src = (const uint4*)lds;
  uint4* out = (uint4*)dst;
  const uint32_t total = valid_threads * (NW / 4);
  for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) out[i] = src[i];
  __syncthreads();
}

// record i = decimal text of the 32-byte little-endian integer i of `src` (a [row][proof] array of
// scalars, or of points, whose x and y are consecutive integers)
__global__ void __launch_bounds__(256) k_dec_numbers(const uint8_t* __restrict__ src, uint32_t count,
                                                    uint8_t* __restrict__ dst) {
  __shared__ uint32_t lds[256 * DEC_REC_WORDS];
  const uint32_t first = blockIdx.x * 256u;
  const uint32_t i = first + threadIdx.x;
  uint32_t w[DEC_REC_WORDS];
#pragma unroll
  for (int j = 0; j < DEC_REC_WORDS; j++) w[j] = 0;
  if (i < count) u256_to_dec_record(ld_u256(src + (size_t)i * 32), w);
  const uint32_t valid = count - first < 256u ? count - first : 256u;
  block_store_records<DEC_REC_WORDS>(lds, w, dst + (size_t)first * DEC_REC, valid);
}

// x and y records of the block's affine results (identity = "0", "0"), to txt[p][2][80]
__device__ __forceinline__ void block_store_point_text(uint32_t* lds, const affine& a, uint8_t* txt, uint32_t B) {
  uint32_t w[2 * DEC_REC_WORDS];
  {
    uint32_t wx[DEC_REC_WORDS], wy[DEC_REC_WORDS];
    u256_to_dec_record(a.x, wx);
    u256_to_dec_record(a.y, wy);
#pragma unroll
    for (int j = 0; j < DEC_REC_WORDS; j++) {
      w[j] = wx[j];
      w[DEC_REC_WORDS + j] = wy[j];
    }
  }
  const uint32_t first = blockIdx.x * blockDim.x;
  const uint32_t valid = B - first < blockDim.x ? B - first : blockDim.x;
  block_store_records<2 * DEC_REC_WORDS>(lds, w, txt + (size_t)first * 2 * DEC_REC, valid);
}

// ---- fixed-base multi-exponentiation: acc[p] = sum_terms k[term][p] * base(term) ---------
// lpp lanes cooperate on one proof (terms strided over them) and are summed by shuffles.
// The table entry of the next non-zero digit is fetched before the current mixed add is
// issued, so the (HBM-resident for 12/16-bit windows) gather overlaps the field arithmetic.
template <int WB>
__global__ void __launch_bounds__(256) k_fixed_multiexp(uint32_t B, uint32_t nterms, uint32_t lpp,
                                                       const uint32_t* __restrict__ dig,
                                                       const uint32_t* __restrict__ term_base,
                                                       const uint8_t* __restrict__ table, xyzz* __restrict__ acc_out) {
  constexpr int W = fb_windows(WB), E = fb_entries(WB), DW = fb_dwords(WB), SLOT = fb_slot_bits(WB), PER = 32 / SLOT;
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t p = t / lpp, sub = t % lpp;
  bool live = p < B;
  xyzz acc = xyzz_inf();
  if (live) {
    // The lpp lanes of a proof share its nterms x W (term, window) digits round-robin: lane `sub` takes the
    // flat indices sub, sub + lpp, ... (splitting by whole terms left up to lpp - 1 term slots idle per launch).
    const uint32_t nf = nterms * (uint32_t)W;
    // Four stages in flight per lane.  For the lane's indices i0 < i1 < i2 < i3 (stride lpp), one loop step
    //   loads the digit word and table base of i3 (their arithmetic waits for the next step),
    //   decodes i2 (address of its table entry),
    //   gathers the entry of i1 (decoded one step ago), and
    //   adds the entry of i0 (gathered one step ago).
    // Every load is unconditional and feeds nothing in its own step, so the waits land behind the mixed addition.
    // (Touching the entry of i2 ahead of its gather - an atomic add of zero, no destination register - was measured:
    // the extra random accesses cost more than the shorter waits save, 6.22 -> 7.16 ms per launch.)
    auto fetch = [&](uint32_t fi, uint32_t& word, uint32_t& tb) {  // loads only; fi < nf
      uint32_t term = fi / (uint32_t)W, w = fi - term * (uint32_t)W;
      word = dig[((size_t)term * DW + w / (uint32_t)PER) * B + p];
      tb = term_base[term];
    };
    struct Dec {
      const uint8_t* addr;
      bool neg, have;
    };
    auto decode = [&](uint32_t fi, uint32_t word, uint32_t tb) {  // fi < nf
      const uint32_t term = fi / (uint32_t)W, w = fi - term * (uint32_t)W, b = w % (uint32_t)PER;
      const uint32_t code = SLOT == 32 ? word : ((word >> ((SLOT * b) & 31)) & (uint32_t)((1ull << SLOT) - 1ull));
      Dec d;
      d.neg = false;
      uint32_t mag = 1;
      d.have = code != 0;
      if (d.have) mag = fb_decode<WB>(code, d.neg);
      d.addr = table + (((size_t)tb * W + w) * E + (mag - 1)) * 64;  // a zero digit reads a valid entry and drops it
      return d;
    };
    affine p0 = affine_inf();  // gathered, not yet added
    bool n0 = false, have0 = false;
    uint32_t f = sub, word_next = 0, tb_next = 0;
    Dec cur;
    cur.addr = table;
    cur.neg = false;
    cur.have = false;
    if (f < nf) {
      fetch(f, word_next, tb_next);
      cur = decode(f, word_next, tb_next);
      fetch(f + lpp < nf ? f + lpp : nf - 1, word_next, tb_next);
    }
    // the accumulator lives on nine 29-bit limbs (fq29.h); ONE addition site
    xyzz29 acc29 = xyzz29_inf();
#pragma unroll 1
    while (f < nf || have0) {
      const bool in_range = f < nf;
      // gather the current index's entry (a drain step re-reads the last one and drops it)
      const affine qn = ld_affine(cur.addr);
      const bool negn = cur.neg, have = in_range && cur.have;
      // decode the next index; load the raw digits of the one after
      const uint32_t f1 = f + lpp < nf ? f + lpp : nf - 1, f2 = f + 2 * lpp < nf ? f + 2 * lpp : nf - 1;
      const Dec nxt = decode(f1, word_next, tb_next);
      fetch(f2, word_next, tb_next);
      // add the entry gathered in the previous step
      if (have0) acc29 = xyzz29_madd(acc29, p0, n0);
      // the new entry becomes the pending one
      p0 = qn;
      n0 = negn;
      have0 = have;
      cur = nxt;
      if (in_range) f += lpp;
    }
    acc = xyzz29_to(acc29);
  }
  for (uint32_t dlt = lpp >> 1; dlt > 0; dlt >>= 1) {
    xyzz o = shfl_down_xyzz(acc, dlt);
    acc = xyzz_add(acc, o);
  }
  if (live && sub == 0) st_xyzz(&acc_out[p], acc);
}

// Small batches: the (term, window) digits of a proof strided over gridDim.x workgroups of 256 lanes, a handful of
// mixed additions per lane; workgroup sums into part[p][block] (or straight into acc_out when one workgroup
// covers the proof), k_fixed_sum adds the workgroups.
template <int WB>
__global__ void __launch_bounds__(256) k_fixed_flat(uint32_t B, uint32_t nterms, const uint32_t* __restrict__ dig,
                                                   const uint32_t* __restrict__ term_base, const uint8_t* __restrict__ table,
                                                   xyzz* __restrict__ part /*[B][gridDim.x]*/, xyzz* __restrict__ acc_out) {
  constexpr int W = fb_windows(WB), E = fb_entries(WB), DW = fb_dwords(WB), SLOT = fb_slot_bits(WB), PER = 32 / SLOT;
  __shared__ uint32_t red[256 * 36];
  const uint32_t p = blockIdx.y, nf = nterms * (uint32_t)W;
  xyzz29 acc = xyzz29_inf();
#pragma unroll 1
  for (uint32_t f = blockIdx.x * 256u + threadIdx.x; f < nf; f += gridDim.x * 256u) {
    const uint32_t term = f / (uint32_t)W, w = f - term * (uint32_t)W;
    uint32_t word = dig[((size_t)term * DW + (w / PER)) * B + p];
    uint32_t code = SLOT == 32 ? word : ((word >> ((SLOT * (w % PER)) & 31)) & (uint32_t)((1ull << SLOT) - 1ull));
    if (code == 0) continue;
    bool neg;
    uint32_t mag = fb_decode<WB>(code, neg);
    acc = xyzz29_madd(acc, ld_affine(table + (((size_t)term_base[term] * W + w) * E + (mag - 1)) * 64), neg);
  }
  xyzz sum = block_sum_quad(xyzz29_to(acc), red);
  if (threadIdx.x == 0) st_xyzz(gridDim.x == 1 ? &acc_out[p] : &part[(size_t)p * gridDim.x + blockIdx.x], sum);
}
__global__ void __launch_bounds__(64) k_fixed_sum(uint32_t nblk /* <= 64 */, const xyzz* __restrict__ part, xyzz* __restrict__ acc_out) {
  __shared__ uint32_t red[64 * 36];
  const uint32_t p = blockIdx.x;
  xyzz acc = xyzz_inf();
  if (threadIdx.x < nblk) acc = ld_xyzz(&part[(size_t)p * nblk + threadIdx.x]);
  acc = block_sum_quad(acc, red);
  if (threadIdx.x == 0) st_xyzz(&acc_out[p], acc);
}

// Joint tables, built before any challenge is known: for a pair of the proof's points (P0, P1) the 15 sums over the
// non-empty subsets of {P0, phi(P0), P1, phi(P1)} as AFFINE points, rtab[(15 t + e - 1) * B + p] (64 B each).
// glv_split_pos leaves no signs, so the tables depend on the proof's points only: this kernel runs on the
// side stream off the host's critical path, and the chains add table entries with mixed additions (8M + 2S)
// instead of building an XYZZ table per use and adding its entries with full additions (12M + 2S).
// Tables 0..k-1 are the rounds' (L_r, R_r); table k is S alone (stage 1's x * S); tables k+1.. pair up the terms of
// the t-polynomial check, V_0..V_{m-1}, T1, T2 (the last one holds a single point when m + 2 is odd, i.e. m = 1).
// A table of a single point has the entries 1..3 only.
// The composite entries are formed in XYZZ on 29-bit limbs (rtx) and normalised with ONE inversion per proof
// (Montgomery's trick over all composite entries; rpre holds the running products).
__device__ __forceinline__ int composite_index(int e, int top) { return e - 2 - top; }  // 3,5,6,7,9..15 -> 0..10

__host__ __device__ inline uint32_t vt_terms(const VDims& d) { return d.m + 2; }                    // V_j, T1, T2
__host__ __device__ inline uint32_t vt_tables(const VDims& d) { return (vt_terms(d) + 1) / 2; }     // their pair tables
__host__ __device__ inline uint32_t vt_first(const VDims& d) { return d.k + 1; }                    // index of the first
__host__ __device__ inline uint32_t joint_tables(const VDims& d) { return d.k + 1 + vt_tables(d); }
// point row of term j of the t-check's list
__device__ __forceinline__ uint32_t vt_row(const VDims& d, uint32_t term) {
  return term < d.m ? VP_L0 + 2 * d.k + term : (term == d.m ? VP_T1 : VP_T2);
}
// rows of the two points of table t; rb = -1: a single point
__device__ __forceinline__ void joint_table_rows(const VDims& d, uint32_t t, int& ra, int& rb) {
  if (t < d.k) {
    ra = (int)(VP_L0 + t);
    rb = (int)(VP_L0 + d.k + t);
  } else if (t == d.k) {
    ra = (int)VP_S;
    rb = -1;
  } else {
    const uint32_t j = 2 * (t - d.k - 1);
    ra = (int)vt_row(d, j);
    rb = j + 1 < vt_terms(d) ? (int)vt_row(d, j + 1) : -1;
  }
}

// limb-major staging of k_round_tables (word j of entry `slot` of proof p at (slot * words + j) * B + p: a wave's
// accesses coalesce)
__device__ __forceinline__ void st_fe29_soa(uint32_t* a, size_t B, const fe29& v) {
#pragma unroll
  for (int i = 0; i < 9; i++) a[(size_t)i * B] = v.v[i];
}
__device__ __forceinline__ fe29 ld_fe29_soa(const uint32_t* a, size_t B) {
  fe29 v;
#pragma unroll
  for (int i = 0; i < 9; i++) v.v[i] = a[(size_t)i * B];
  return v;
}
__device__ __forceinline__ void st_xyzz29_soa(uint32_t* a, size_t B, const xyzz29& v) {
  st_fe29_soa(a, B, v.X);
  st_fe29_soa(a + 9 * B, B, v.Y);
  st_fe29_soa(a + 18 * B, B, v.ZZ);
  st_fe29_soa(a + 27 * B, B, v.ZZZ);
}
__device__ __forceinline__ xyzz29 ld_xyzz29_soa(const uint32_t* a, size_t B) {
  xyzz29 v;
  v.X = ld_fe29_soa(a, B);
  v.Y = ld_fe29_soa(a + 9 * B, B);
  v.ZZ = ld_fe29_soa(a + 18 * B, B);
  v.ZZZ = ld_fe29_soa(a + 27 * B, B);
  return v;
}

// The composite entries are sums on the nine 29-bit limbs of fq29.h (one mixed addition each, ~1,600 instructions
// against ~2,300 on 8 x 32 limbs; Montgomery's trick and the normalisation on the same limbs).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_round_tables(VDims d, const uint8_t* __restrict__ pts, uint32_t* __restrict__ rtx /*[11 T][36][B]*/,
               uint32_t* __restrict__ rpre /*[11 T][9][B]*/, uint8_t* __restrict__ rtab /*[15 T][B][64]*/,
               uint32_t t_lo, uint32_t t_hi, uint32_t t_skip /*tables t_lo <= t < t_hi except t_skip*/) {
  __shared__ u256 inv_lds[BATCHINV_LDS];
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p < d.B;
  const size_t B = d.B, ES = B * 64;  // stride between table entries of one proof
  fe29 run = fe29_zero();
  run.v[0] = 1;
  bool need_inv = false;  // uniform: some table of the range holds two points (a single point's table is inversion-free)
#pragma unroll 1
  for (uint32_t r = t_lo; r < t_hi; r++) {
    int ra, rb;
    joint_table_rows(d, r, ra, rb);
    need_inv = need_inv || (r != t_skip && rb >= 0);
  }
#pragma unroll 1
  for (uint32_t r = t_lo; live && r < t_hi; r++) {
    if (r == t_skip) continue;
    uint8_t* T = rtab + ((size_t)r * 15 * B + p) * 64;
    int ra, rb;
    joint_table_rows(d, r, ra, rb);
    affine L = ld_affine(pts + ((size_t)ra * B + p) * 64);
    st_affine(T + 0 * ES, L);
    st_affine(T + 1 * ES, glv_phi(L));
    st_affine(T + 2 * ES, glv_p_plus_phi(L));  // entry 3 = P0 + phi(P0), no addition needed
    if (rb < 0) continue;
    affine Rp = ld_affine(pts + ((size_t)rb * B + p) * 64);
    st_affine(T + 3 * ES, Rp);
    st_affine(T + 7 * ES, glv_phi(Rp));
    st_affine(T + 11 * ES, glv_p_plus_phi(Rp));  // entry 12 = P1 + phi(P1)
#pragma unroll 1
    for (int e = 5; e < 16; e++) {
      int top = 31 - __builtin_clz((unsigned)e);
      int rest = e & ~(1 << top);
      if (rest == 0 || e == 12) continue;
      affine qt = ld_affine(T + (size_t)((1 << top) - 1) * ES);
      xyzz29 base;
      if ((rest & (rest - 1)) == 0 || rest == 3) {  // an entry that is affine already
        base = xyzz29_from_affine(ld_affine(T + (size_t)(rest - 1) * ES), false);
      } else {
        int rtop = 31 - __builtin_clz((unsigned)rest);
        base = ld_xyzz29_soa(rtx + ((size_t)r * 11 + composite_index(rest, rtop)) * 36 * B + p, B);
      }
      // xyzz29_madd wants a finite entry: O (an L_r or R_r at infinity) leaves the sum as it is
      const xyzz29 sum = xyzz29_madd(base, qt, false);
      const xyzz29 v = affine_is_inf(qt) ? base : sum;
      const size_t slot = (size_t)r * 11 + composite_index(e, top);
      st_xyzz29_soa(rtx + slot * 36 * B + p, B, v);
      st_fe29_soa(rpre + slot * 9 * B + p, B, run);
      if (!xyzz29_is_inf(v)) run = fe29_mul(run, v.ZZZ);
    }
  }
  if (!need_inv) return;
  // run is a product of non-zero ZZZ values; the workgroup shares one inversion (batchinv.h)
  fe29 inv = fe29_from_u256(block_batch_inverse<FqInvOps>(fe29_to_u256(run), inv_lds));
  if (!live) return;
#pragma unroll 1
  for (uint32_t rr = t_hi; rr-- > t_lo;) {
    if (rr == t_skip) continue;
    uint8_t* T = rtab + ((size_t)rr * 15 * B + p) * 64;
    int ra, rb;
    joint_table_rows(d, rr, ra, rb);
    if (rb < 0) continue;
#pragma unroll 1
    for (int e = 15; e >= 5; e--) {
      int top = 31 - __builtin_clz((unsigned)e);
      if ((e & ~(1 << top)) == 0 || e == 12) continue;
      const size_t slot = (size_t)rr * 11 + composite_index(e, top);
      const xyzz29 v = ld_xyzz29_soa(rtx + slot * 36 * B + p, B);
      affine a = affine_inf();
      if (!xyzz29_is_inf(v)) {
        const fe29 zi = fe29_mul(inv, ld_fe29_soa(rpre + slot * 9 * B + p, B));  // 1 / ZZZ of this entry
        inv = fe29_mul(inv, v.ZZZ);
        const fe29 zz_inv = fe29_sqr(fe29_mul(v.ZZ, zi));  // 1 / ZZ = (ZZ / ZZZ)^2
        a.x = fq_canon(fe29_to_u256(fe29_mul(v.X, zz_inv)));
        a.y = fq_canon(fe29_to_u256(fe29_mul(v.Y, zi)));
      }
      st_affine(T + (size_t)(e - 1) * ES, a);
    }
  }
}

// stage 1 tail: P = acc1 + x*S + A  -> affine (kept on device and exported).  x * S runs on S's joint table (table k of
// k_round_tables, complete before this kernel is launched): a doubling and at most one mixed addition per bit.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_stage1_tail(VDims d, const uint8_t* __restrict__ pts /*[..][B][64]*/,
                                                    const u256* __restrict__ cst, const xyzz* __restrict__ acc1,
                                                    const uint8_t* __restrict__ rtab, uint8_t* __restrict__ acc_aff /*[B][64]*/,
                                                    uint8_t* __restrict__ host_out /*pinned host copy, or null*/,
                                                    uint8_t* __restrict__ host_txt /*pinned [B][2][80] decimal text, or null*/) {
  __shared__ uint32_t lds[256 * 2 * DEC_REC_WORDS];
  // the host waits for this kernel (it hashes P next): its waves take issue priority over the throughput
  // kernels of other chunks that share the SIMDs
  __builtin_amdgcn_s_setprio(3);
  __shared__ u256 inv_lds[BATCHINV_LDS];
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p < d.B;
  xyzz acc = xyzz_inf();
  if (live) {
    size_t B = d.B;
    const u256 k[2] = {Fr::from_mont(ldm(cst, C_X * B + p)), u256_zero()};
    const bool present[2] = {true, false};
    const uint8_t* T = rtab + ((size_t)d.k * 15 * B + p) * 64;
    acc = glv_tables_multiexp29<1>(k, present, [&](int, int e) { return ld_affine(T + (size_t)(e - 1) * B * 64); });
    acc = xyzz_add(acc, ld_xyzz(&acc1[p]));
    acc = xyzz_madd(acc, ld_affine(pts + ((size_t)VP_A * B + p) * 64));
  }
  // affine form with one inversion per workgroup; the identity (ZZZ = 0) comes back as 0 and stays (0, 0)
  u256 zi = block_batch_inverse<FqInvOps>(live ? acc.ZZZ : u256_one(), inv_lds);
  affine out = affine_inf();
  if (live) {
    out = xyzz_to_affine_with_inv(acc, zi);
    st_affine(acc_aff + (size_t)p * 64, out);
    if (host_out) st_affine(host_out + (size_t)p * 64, out);
  }
  if (host_txt) block_store_point_text(lds, out, host_txt, d.B);  // uniform branch: every thread of the block arrives
}

// one IPP round of mkChallenges: acc <- x^2 L + x^-2 R + acc  (Verifier.hs:47-50)
template <bool BATCH_INV>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_round(VDims d, uint32_t round,
                                              const uint8_t* __restrict__ xk /*[B][32] canonical*/,
                                              u256* __restrict__ xk_m /*[k][B]: x_k^2 */, u256* __restrict__ xkinv_m,
                                              const uint8_t* __restrict__ rtab, uint8_t* __restrict__ acc_aff,
                                              uint8_t* __restrict__ host_out /*pinned host copy, or null*/,
                                              uint8_t* __restrict__ host_txt /*pinned [B][2][80] decimal text, or null*/) {
  __shared__ uint32_t lds[256 * 2 * DEC_REC_WORDS];
  __builtin_amdgcn_s_setprio(3);  // on the host's critical path, like k_stage1_tail
  __shared__ u256 inv_lds[BATCHINV_LDS];
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p < d.B;
  const size_t B = d.B;
  u256 x = Fr::one();
  if (live) x = Fr::to_mont(Fr::reduce_once(ld_u256(xk + (size_t)p * 32)));
  // x^-1 and, below, the affine form of the result: one inversion per workgroup each (batchinv.h)
  u256 xinv = BATCH_INV ? block_batch_inverse<FrInvOps>(x, inv_lds) : fr_inv_gcd(x);
  xyzz acc = xyzz_inf();
  if (live) {
    const u256 x2 = Fr::sqr(x);
    stm(xk_m, (size_t)round * B + p, x2);  // mkOtherExponents needs the squares only (Verifier.hs:66-71)
    stm(xkinv_m, (size_t)round * B + p, xinv);
    const uint8_t* T = rtab + ((size_t)round * 15 * B + p) * 64;
    acc = glv_pair_multiexp_tab29(Fr::from_mont(x2), Fr::from_mont(Fr::sqr(xinv)),
                                [&](int e) { return ld_affine(T + (size_t)(e - 1) * B * 64); });
    acc = xyzz_madd(acc, ld_affine(acc_aff + (size_t)p * 64));
  }
  const u256 zzz = live ? acc.ZZZ : u256_one();
  u256 zi = BATCH_INV ? block_batch_inverse<FqInvOps>(zzz, inv_lds) : fq_inv_gcd(zzz);
  affine out = affine_inf();
  if (live) {
    out = xyzz_to_affine_with_inv(acc, zi);
    st_affine(acc_aff + (size_t)p * 64, out);
    if (host_out) st_affine(host_out + (size_t)p * 64, out);
  }
  if (host_txt) block_store_point_text(lds, out, host_txt, d.B);
}

// The same round with the 129-step chain of every proof spread over a PAIR of lanes (lanes29.h): a doubling and a mixed
// addition are 5 + 5 rounds of two independent products instead of 9 + 10 dependent ones, which is what the host waits
// for six times per chunk.  A workgroup of 256 lanes serves 128 proofs: lanes 0..127 (two waves) do the scalar work of
// one proof each (x^2, x^-1, the GLV halves), hand the half-scalars to the pairs through LDS, all 256 lanes run the
// chains, and lanes 0..127 take the sums back for the running commitment, its affine form and its text.  The other two
// waves skip the scalar parts as whole waves, so those cost no more instructions than in k_round.
constexpr uint32_t RP_PROOFS = 128;
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_round_pair(VDims d, uint32_t round, const uint8_t* __restrict__ xk /*[B][32] canonical*/, u256* __restrict__ xk_m,
             u256* __restrict__ xkinv_m, const uint8_t* __restrict__ rtab, uint8_t* __restrict__ acc_aff,
             uint8_t* __restrict__ host_out, uint8_t* __restrict__ host_txt) {
  __shared__ uint32_t lds[256 * 2 * DEC_REC_WORDS];
  __shared__ uint32_t xch[RP_PROOFS * 36];  // per proof: 4 x 5 half-scalar words on the way in, X, Y, ZZ, ZZZ on the way out
  __builtin_amdgcn_s_setprio(3);
  const uint32_t tid = threadIdx.x, first = blockIdx.x * RP_PROOFS;
  const uint32_t p = first + tid;
  const bool solo = tid < RP_PROOFS, live = solo && p < d.B;
  const size_t B = d.B;
  if (solo) {  // whole waves
    u256 x = Fr::one();
    if (live) x = Fr::to_mont(Fr::reduce_once(ld_u256(xk + (size_t)p * 32)));
    const u256 xinv = fr_inv_gcd(x);
    GlvPos s0, s1;
#pragma unroll
    for (int i = 0; i < 5; i++) s0.m1[i] = s0.m2[i] = s1.m1[i] = s1.m2[i] = 0;
    if (live) {
      const u256 x2 = Fr::sqr(x);
      stm(xk_m, (size_t)round * B + p, x2);
      stm(xkinv_m, (size_t)round * B + p, xinv);
      s0 = glv_split_pos(Fr::from_mont(x2));
      s1 = glv_split_pos(Fr::from_mont(Fr::sqr(xinv)));
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
      xch[tid * 36 + i] = s0.m1[i];
      xch[tid * 36 + 5 + i] = s0.m2[i];
      xch[tid * 36 + 10 + i] = s1.m1[i];
      xch[tid * 36 + 15 + i] = s1.m2[i];
    }
  }
  __syncthreads();
  {
    const uint32_t q = tid >> 1, pq = first + q;
    const bool odd = (tid & 1u) != 0;
    uint32_t m[4][5];
#pragma unroll
    for (int h = 0; h < 4; h++)
#pragma unroll
      for (int i = 0; i < 5; i++) m[h][i] = xch[q * 36 + h * 5 + i];
    __syncthreads();  // every pair has its half-scalars: the area is free for the results
    pair29 acc;
    acc.c0 = fe29_zero();
    acc.c1 = fe29_zero();
    if (pq < d.B) {  // both lanes of a pair agree
      BitReg b0 = bitreg_init5(m[0]), b1 = bitreg_init5(m[1]), b2 = bitreg_init5(m[2]), b3 = bitreg_init5(m[3]);
      const uint8_t* T = rtab + ((size_t)round * 15 * B + pq) * 64;
#pragma unroll 1
      for (int i = 0; i < GLV_BITS; i++) {
        const uint32_t idx = bitreg_next(b0) | (bitreg_next(b1) << 1) | (bitreg_next(b2) << 2) | (bitreg_next(b3) << 3);
        acc = pair_dbl(acc, odd);
        if (idx) acc = pair_madd(acc, T + (size_t)(idx - 1) * B * 64, odd);
      }
    }
    // X, Y, ZZ, ZZZ of proof q at xch[36 q + 0 / 9 / 18 / 27]
#pragma unroll
    for (int i = 0; i < 9; i++) {
      xch[q * 36 + (odd ? 0 : 9) + i] = acc.c0.v[i];
      xch[q * 36 + (odd ? 18 : 27) + i] = acc.c1.v[i];
    }
  }
  __syncthreads();
  affine out = affine_inf();
  if (live) {
    xyzz29 a29;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      a29.X.v[i] = xch[tid * 36 + i];
      a29.Y.v[i] = xch[tid * 36 + 9 + i];
      a29.ZZ.v[i] = xch[tid * 36 + 18 + i];
      a29.ZZZ.v[i] = xch[tid * 36 + 27 + i];
    }
    xyzz acc = xyzz_madd(xyzz29_to(a29), ld_affine(acc_aff + (size_t)p * 64));
    out = xyzz_to_affine_with_inv(acc, fq_inv_gcd(acc.ZZZ));
    st_affine(acc_aff + (size_t)p * 64, out);
    if (host_out) st_affine(host_out + (size_t)p * 64, out);
  }
  if (host_txt) {  // uniform: every lane of the workgroup arrives; lanes 128.. hold empty records
    uint32_t w[2 * DEC_REC_WORDS];
#pragma unroll
    for (int j = 0; j < 2 * DEC_REC_WORDS; j++) w[j] = 0;
    if (solo) {
      uint32_t wx[DEC_REC_WORDS], wy[DEC_REC_WORDS];
      u256_to_dec_record(out.x, wx);
      u256_to_dec_record(out.y, wy);
#pragma unroll
      for (int j = 0; j < DEC_REC_WORDS; j++) {
        w[j] = wx[j];
        w[DEC_REC_WORDS + j] = wy[j];
      }
    }
    const uint32_t valid = d.B - first < RP_PROOFS ? d.B - first : RP_PROOFS;
    block_store_records<2 * DEC_REC_WORDS>(lds, w, host_txt + (size_t)first * 2 * DEC_REC, valid);
  }
}

// The stage-1 tail in the same form: P = acc1 + x*S + A with the chain of x*S on a pair of lanes.  x = m1 + m2 lambda with
// non-negative halves (glv_split_pos), so the chain adds the AFFINE points S and phi(S) (both free) by mixed additions -
// one per set bit, 1.0 per step on average - instead of full additions of an XYZZ table entry.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_stage1_tail_pair(VDims d, const uint8_t* __restrict__ pts, const u256* __restrict__ cst, const xyzz* __restrict__ acc1,
                   uint8_t* __restrict__ sphi /*[B][2][64]: S, phi(S)*/, uint8_t* __restrict__ acc_aff,
                   uint8_t* __restrict__ host_out, uint8_t* __restrict__ host_txt) {
  __shared__ uint32_t lds[256 * 2 * DEC_REC_WORDS];
  __shared__ uint32_t xch[RP_PROOFS * 36];
  __builtin_amdgcn_s_setprio(3);
  const uint32_t tid = threadIdx.x, first = blockIdx.x * RP_PROOFS;
  const uint32_t p = first + tid;
  const bool solo = tid < RP_PROOFS, live = solo && p < d.B;
  const size_t B = d.B;
  if (solo) {  // whole waves
    GlvPos s0;
#pragma unroll
    for (int i = 0; i < 5; i++) s0.m1[i] = s0.m2[i] = 0;
    if (live) {
      s0 = glv_split_pos(Fr::from_mont(ldm(cst, C_X * B + p)));
      const affine S = ld_affine(pts + ((size_t)VP_S * B + p) * 64);
      st_affine(sphi + (size_t)p * 128, S);
      st_affine(sphi + (size_t)p * 128 + 64, glv_phi(S));
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
      xch[tid * 36 + i] = s0.m1[i];
      xch[tid * 36 + 5 + i] = s0.m2[i];
    }
  }
  __syncthreads();  // also orders the workgroup's stores to sphi before its loads below
  {
    const uint32_t q = tid >> 1, pq = first + q;
    const bool odd = (tid & 1u) != 0;
    uint32_t m[2][5];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int i = 0; i < 5; i++) m[h][i] = xch[q * 36 + h * 5 + i];
    __syncthreads();
    pair29 acc;
    acc.c0 = fe29_zero();
    acc.c1 = fe29_zero();
    if (pq < d.B) {
      BitReg b0 = bitreg_init5(m[0]), b1 = bitreg_init5(m[1]);
      const uint8_t* E = sphi + (size_t)pq * 128;
#pragma unroll 1
      for (int i = 0; i < GLV_BITS; i++) {
        const uint32_t idx = bitreg_next(b0) | (bitreg_next(b1) << 1);
        acc = pair_dbl(acc, odd);
#pragma unroll 1
        for (uint32_t t = 0; t < 2; t++)  // ONE addition site
          if ((idx >> t) & 1u) acc = pair_madd(acc, E + t * 64, odd);
      }
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
      xch[q * 36 + (odd ? 0 : 9) + i] = acc.c0.v[i];
      xch[q * 36 + (odd ? 18 : 27) + i] = acc.c1.v[i];
    }
  }
  __syncthreads();
  affine out = affine_inf();
  if (live) {
    xyzz29 a29;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      a29.X.v[i] = xch[tid * 36 + i];
      a29.Y.v[i] = xch[tid * 36 + 9 + i];
      a29.ZZ.v[i] = xch[tid * 36 + 18 + i];
      a29.ZZZ.v[i] = xch[tid * 36 + 27 + i];
    }
    xyzz acc = xyzz_add(xyzz29_to(a29), ld_xyzz(&acc1[p]));
    acc = xyzz_madd(acc, ld_affine(pts + ((size_t)VP_A * B + p) * 64));
    out = xyzz_to_affine_with_inv(acc, fq_inv_gcd(acc.ZZZ));
    st_affine(acc_aff + (size_t)p * 64, out);
    if (host_out) st_affine(host_out + (size_t)p * 64, out);
  }
  if (host_txt) {
    uint32_t w[2 * DEC_REC_WORDS];
#pragma unroll
    for (int j = 0; j < 2 * DEC_REC_WORDS; j++) w[j] = 0;
    if (solo) {
      uint32_t wx[DEC_REC_WORDS], wy[DEC_REC_WORDS];
      u256_to_dec_record(out.x, wx);
      u256_to_dec_record(out.y, wy);
#pragma unroll
      for (int j = 0; j < DEC_REC_WORDS; j++) {
        w[j] = wx[j];
        w[DEC_REC_WORDS + j] = wy[j];
      }
    }
    const uint32_t valid = d.B - first < RP_PROOFS ? d.B - first : RP_PROOFS;
    block_store_records<2 * DEC_REC_WORDS>(lds, w, host_txt + (size_t)first * 2 * DEC_REC, valid);
  }
}

// ---- small batches: the latency path ------------------------------------------------------------------------
// With a handful of proofs the 129-step chains of k_round / k_stage1_tail / k_var_terms leave the GPU idle and set
// the latency of a verification.  Every point of a proof (S, T1, T2, L_r, R_r, V_j) is known before any challenge:
// ipp.hip's k_bit_tables writes 2^j * point for j < 128 once per pass (side stream, beside the host's first hashes), and a
// product k * point is then one lane per bit (glv.h, glv_bit_lane: at most one addition) and a workgroup sum.
// The field inversions of the path (y^-1, (y - 1)^-1, x_r^-1, and the normalisation of each running commitment)
// are done by the host between its hashes, microseconds each against ~0.14 ms for one lane of the GPU.
// Measured (bench.py --batch, one chunk; ms per pass, this path against one lane / a pair of lanes per proof):
//   16: 1.23   64: 1.68 / 6.74   128: 1.99 / 7.05   256: 2.65 / 7.00   512: 3.80 / 7.14   1,024: 6.45 / 7.39   2,048: 12.0 / 7.8
// (the host's inversions run on the job's hashing threads; with one thread the paths cross near 100 proofs)
constexpr uint32_t VSPLIT_MAX_B_DEFAULT = 768;
constexpr uint32_t VSPLIT_SERIAL_MAX_B = 96;  // without lent threads (bp_vbatch_* called directly)
static uint32_t vsplit_max_b() {  // BP_VSPLIT_MAX: largest batch that takes this path
  static const uint32_t v = [] {
    const char* e = getenv("BP_VSPLIT_MAX");
    long x = e ? atol(e) : (long)VSPLIT_MAX_B_DEFAULT;
    return (uint32_t)(x < 0 ? 0 : (x > 4096 ? 4096 : x));
  }();
  return v;
}
static void vbatch_host_for(bp_vbatch* vb, size_t n, const std::function<void(size_t, size_t)>& body) {
  if (vb->pfor && n > 8)
    vb->pfor(n, body);
  else
    body(0, n);
}
enum { HI_YINV = 0, HI_YM1INV = 1, HI_XINV = 2 };

// bit-table index of row `row` (>= VP_S) of proof p
__device__ __forceinline__ size_t sh_index(uint32_t row, size_t B, uint32_t p) { return ((size_t)(row - VP_S) * B + p) * GLV_LANE_BITS; }

// thread per proof: k_vprep0 with the host's inversions (hinv) and the closed-form power sum
__global__ void __launch_bounds__(64) k_vprep0_split(VDims d, const uint8_t* __restrict__ chal /*[4][B][32]*/,
                                                    const uint8_t* __restrict__ hinv, u256* __restrict__ cst,
                                                    u256* __restrict__ zpow /*[m+3][B]*/) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.B) return;
  size_t B = d.B;
  u256 y = Fr::to_mont(Fr::reduce_once(ld_u256(chal + ((size_t)0 * B + p) * 32)));
  u256 zp = Fr::reduce_once(ld_u256(chal + ((size_t)1 * B + p) * 32));
  u256 z = Fr::to_mont(zp);
  u256 x = Fr::to_mont(Fr::reduce_once(ld_u256(chal + ((size_t)2 * B + p) * 32)));
  u256 u = Fr::to_mont(Fr::reduce_once(ld_u256(chal + ((size_t)3 * B + p) * 32)));
  stm(cst, C_ZP * B + p, zp);
  stm(cst, C_Y * B + p, y);
  stm(cst, C_Z * B + p, z);
  stm(cst, C_X * B + p, x);
  stm(cst, C_U * B + p, u);
  stm(cst, C_YINV * B + p, ld_u256(hinv + ((size_t)HI_YINV * B + p) * 32));
  stm(cst, C_X2 * B + p, Fr::sqr(x));
  // sum_i y^i, i < nm, in closed form (y^nm - 1) / (y - 1), nm a power of two; y = 1 gives nm (as k_vprep0)
  u256 ynm = y;
  for (uint32_t j = 0; j < d.k; j++) ynm = Fr::sqr(ynm);
  u256 ym1 = Fr::sub(y, Fr::one());
  u256 ysum;
  if (u256_is_zero(ym1)) {
    u256 nmv = u256_zero();
    nmv.v[0] = d.nm;
    ysum = Fr::to_mont(nmv);
  } else {
    ysum = Fr::mul(Fr::sub(ynm, Fr::one()), ld_u256(hinv + ((size_t)HI_YM1INV * B + p) * 32));
  }
  u256 zc = Fr::one();
  for (uint32_t e = 0; e < d.m + 3; e++) {
    stm(zpow, (size_t)e * B + p, zc);
    zc = Fr::mul(zc, z);
  }
  u256 z2 = Fr::sqr(z);
  u256 two_n = u256_zero();
  two_n.v[d.n >> 5] = 1u << (d.n & 31);
  u256 tn1 = Fr::sub(Fr::to_mont(Fr::reduce_once(two_n)), Fr::one());
  u256 zs = u256_zero(), zj = Fr::mul(z2, z);
  for (uint32_t j = 0; j < d.m; j++) {
    zs = Fr::add(zs, zj);
    zj = Fr::mul(zj, z);
  }
  stm(cst, C_DELTA * B + p, Fr::sub(Fr::mul(Fr::sub(z, z2), ysum), Fr::mul(zs, tn1)));
}

// P = acc1 + x*S + A (computeLRCommitment), one workgroup of 128 lanes per proof; XYZZ to acc_x and the pinned copy
__global__ void __launch_bounds__(128) k_stage1_tail_bits(VDims d, const uint8_t* __restrict__ pts, const u256* __restrict__ cst,
                                                         const xyzz* __restrict__ acc1, const xyzz* __restrict__ sh,
                                                         xyzz* __restrict__ acc_x, xyzz* __restrict__ hx) {
  __shared__ uint32_t red[128 * 36];
  const uint32_t p = blockIdx.x;
  const size_t B = d.B;
  xyzz term = bit_lane_term(Fr::from_mont(ldm(cst, C_X * B + p)), sh + sh_index(VP_S, B, p), (int)threadIdx.x);
  xyzz sum = block_sum_quad(term, red);
  if (threadIdx.x == 0) {
    sum = xyzz_add(sum, ld_xyzz(&acc1[p]));
    sum = xyzz_madd(sum, ld_affine(pts + ((size_t)VP_A * B + p) * 64));
    st_xyzz(&acc_x[p], sum);
    st_xyzz(&hx[p], sum);
  }
}

// acc <- x_r^2 L_r + x_r^-2 R_r + acc (InnerProductProof/Verifier.hs:49), 256 lanes per proof: lanes 0..127 the bits
// of x_r^2 on L_r, lanes 128..255 those of x_r^-2 on R_r
__global__ void __launch_bounds__(256) k_round_bits(VDims d, uint32_t round, const uint8_t* __restrict__ xk,
                                                   const uint8_t* __restrict__ hinv, u256* __restrict__ xk_m,
                                                   u256* __restrict__ xkinv_m, const xyzz* __restrict__ sh,
                                                   xyzz* __restrict__ acc_x, xyzz* __restrict__ hx) {
  __shared__ uint32_t red[256 * 36];
  const uint32_t p = blockIdx.x;
  const size_t B = d.B;
  const int t = (int)threadIdx.x, side = t >> 7, j = t & (GLV_LANE_BITS - 1);
  u256 x = Fr::to_mont(Fr::reduce_once(ld_u256(xk + (size_t)p * 32)));
  u256 xinv = ld_u256(hinv + ((size_t)HI_XINV * B + p) * 32);
  u256 x2 = Fr::sqr(x);
  u256 k = Fr::from_mont(side == 0 ? x2 : Fr::sqr(xinv));
  xyzz term = bit_lane_term(k, sh + sh_index(VP_L0 + (side ? d.k : 0) + round, B, p), j);
  xyzz sum = block_sum_quad(term, red);
  if (t == 0) {
    stm(xk_m, (size_t)round * B + p, x2);
    stm(xkinv_m, (size_t)round * B + p, xinv);
    sum = xyzz_add(sum, ld_xyzz(&acc_x[p]));
    st_xyzz(&acc_x[p], sum);
    st_xyzz(&hx[p], sum);
  }
}

// acc2 += sum over the m + 2 variable-base terms of E2 (V_0..V_{m-1}, T1, T2 with the scalars vs of k_vprep_t), 256
// lanes per proof: two terms per pass over the bit positions, one workgroup sum at the end
__global__ void __launch_bounds__(256) k_tcheck_bits(VDims d, const uint8_t* __restrict__ vs, const xyzz* __restrict__ sh,
                                                    xyzz* __restrict__ acc2) {
  __shared__ uint32_t red[256 * 36];
  const uint32_t p = blockIdx.x;
  const size_t B = d.B;
  const int t = (int)threadIdx.x, side = t >> 7, j = t & (GLV_LANE_BITS - 1);
  const uint32_t nv = d.m + 2;
  xyzz acc = xyzz_inf();
#pragma unroll 1
  for (uint32_t term = (uint32_t)side; term < nv; term += 2) {
    uint32_t row = term < d.m ? VP_L0 + 2 * d.k + term : (term == d.m ? VP_T1 : VP_T2);
    acc = xyzz_add(acc, bit_lane_term(ld_u256(vs + ((size_t)term * B + p) * 32), sh + sh_index(row, B, p), j));
  }
  xyzz sum = block_sum_quad(acc, red);
  if (t == 0) st_xyzz(&acc2[p], xyzz_add(sum, ld_xyzz(&acc2[p])));
}

// mkOtherExponents (Verifier.hs:55-73): s_i = s0 * prod_{bit j of i} x_{k-1-j}^2, s0 = prod_k x_k^-1 (challenges are
// listed last round first, so bit 0 pairs with the LAST round).  The index is split into its low kl = k/2 bits
// and the rest: s_i = lo[i & (2^kl - 1)] * hi[i >> kl], with s0 folded into lo.  Thread per proof: both
// half-tables by one multiplication per entry, stab[e * B + p], lo entries first.
__device__ __forceinline__ uint32_t vsplit_lo_bits(uint32_t k) { return k >> 1; }

__global__ void __launch_bounds__(256) k_vprep3a(VDims d, const u256* __restrict__ xk2_m, const u256* __restrict__ xkinv_m,
                                                u256* __restrict__ cst, u256* __restrict__ stab) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.B) return;
  size_t B = d.B;
  u256 s0 = Fr::one();
  for (uint32_t r = 0; r < d.k; r++) s0 = Fr::mul(s0, ldm(xkinv_m, (size_t)r * B + p));
  stm(cst, C_S0 * B + p, s0);
  const uint32_t kl = vsplit_lo_bits(d.k), kh = d.k - kl, nlo = 1u << kl, nhi = 1u << kh;
  stm(stab, (size_t)0 * B + p, s0);
  for (uint32_t e = 1; e < nlo; e++) {
    uint32_t j = (uint32_t)__builtin_ctz(e);
    stm(stab, (size_t)e * B + p,
        Fr::mul(ldm(stab, (size_t)(e & (e - 1)) * B + p), ldm(xk2_m, (size_t)(d.k - 1 - j) * B + p)));
  }
  stm(stab, (size_t)nlo * B + p, Fr::one());
  for (uint32_t e = 1; e < nhi; e++) {
    uint32_t j = (uint32_t)__builtin_ctz(e);
    stm(stab, (size_t)(nlo + e) * B + p,
        Fr::mul(ldm(stab, (size_t)(nlo + (e & (e - 1))) * B + p), ldm(xk2_m, (size_t)(d.k - 1 - (kl + j)) * B + p)));
  }
}

// thread per (i, p): emits
//   term i            (base G_i):        l * s_i
//   term nm + (nm-1-i) (base H_{nm-1-i}): r * s_i * y^-(nm-1-i)        (reverse otherExponents, :38)
// and for i == 0 the remaining single term.  l and r enter in plain form: the Montgomery product of a plain and a
// Montgomery-form operand is the plain product, so the scalars come out canonical without conversions.
__global__ void __launch_bounds__(256) k_vprep3(VDims d, const uint8_t* __restrict__ sc, const u256* __restrict__ cst,
                                               const u256* __restrict__ ypowinv, const u256* __restrict__ stab,
                                               uint32_t* __restrict__ dig_a) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t B = d.B;
  if (t >= (size_t)d.nm * B) return;
  uint32_t i = (uint32_t)(t / B), p = (uint32_t)(t % B);
  const uint32_t kl = vsplit_lo_bits(d.k), nlo = 1u << kl;
  u256 s = Fr::mul(ldm(stab, (size_t)(i & (nlo - 1)) * B + p), ldm(stab, (size_t)(nlo + (i >> kl)) * B + p));
  u256 lp = Fr::reduce_once(ld_u256(sc + ((size_t)3 * B + p) * 32));
  u256 rp = Fr::reduce_once(ld_u256(sc + ((size_t)4 * B + p) * 32));
  put_digits(dig_a, i, d, p, Fr::mul(lp, s));
  uint32_t hi = d.nm - 1 - i;
  put_digits(dig_a, d.nm + hi, d, p, Fr::mul(Fr::mul(rp, s), ldm(ypowinv, (size_t)hi * B + p)));
  if (i == 0) {
    u256 u = ldm(cst, C_U * B + p);
    put_digits(dig_a, 2 * d.nm, d, p, Fr::mul(Fr::mul(lp, Fr::to_mont(rp)), u));  // (l r u) G
  }
}

// thread per proof: scalars of the t-polynomial check E2 = (t - delta) G + tau H - sum_j z^(j+2) V_j - x T1 - x^2 T2
// (verifyTPoly, MultiRangeProof/Verifier.hs:43-65).  They depend on y, z, x only, so this check runs on a side
// stream while the host is busy with the rounds.
__global__ void __launch_bounds__(256) k_vprep_t(VDims d, const uint8_t* __restrict__ sc, const u256* __restrict__ cst,
                                                const u256* __restrict__ zpow, uint32_t* __restrict__ dig_b,
                                                uint8_t* __restrict__ vs /*[m+2][B][32]*/) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.B) return;
  size_t B = d.B;
  u256 tt = Fr::to_mont(Fr::reduce_once(ld_u256(sc + ((size_t)2 * B + p) * 32)));
  u256 tau = Fr::reduce_once(ld_u256(sc + ((size_t)0 * B + p) * 32));
  put_digits(dig_b, 0, d, p, Fr::from_mont(Fr::sub(tt, ldm(cst, C_DELTA * B + p))));  // (t - delta) G
  put_digits(dig_b, 1, d, p, tau);                                                    // tau H
  for (uint32_t j = 0; j < d.m; j++)
    st_u256(vs + ((size_t)j * B + p) * 32, Fr::from_mont(Fr::neg(ldm(zpow, (size_t)(j + 2) * B + p))));
  st_u256(vs + ((size_t)d.m * B + p) * 32, Fr::from_mont(Fr::neg(ldm(cst, C_X * B + p))));
  st_u256(vs + ((size_t)(d.m + 1) * B + p) * 32, Fr::from_mont(Fr::neg(ldm(cst, C_X2 * B + p))));
}

// acc2 += the terms of NT joint tables of the t-check's list (V_0..V_{m-1}, T1, T2 in pairs, tables vt_first + t0 ..
// of k_round_tables): up to four terms on one doubling chain.
template <int NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k_var_terms(VDims d, uint32_t t0, const uint8_t* __restrict__ vs,
                                                  const uint8_t* __restrict__ rtab, xyzz* __restrict__ acc2) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.B) return;
  size_t B = d.B;
  u256 k[2 * NT];
  bool present[2 * NT];
#pragma unroll
  for (int j = 0; j < 2 * NT; j++) {
    const uint32_t term = 2 * t0 + j;
    present[j] = term < vt_terms(d);
    k[j] = present[j] ? ld_u256(vs + ((size_t)term * B + p) * 32) : u256_zero();
  }
  const uint8_t* T = rtab + ((size_t)(vt_first(d) + t0) * 15 * B + p) * 64;
  xyzz acc = glv_tables_multiexp29<NT>(k, present, [&](int t, int e) {
    return ld_affine(T + ((size_t)t * 15 + (size_t)(e - 1)) * B * 64);
  });
  acc = xyzz_add(acc, ld_xyzz(&acc2[p]));
  st_xyzz(&acc2[p], acc);
}

// ---- the transcript on the device (fsdev.h): challenges formed where the proofs and the running commitment live -----
__device__ __forceinline__ FsPoint ld_fspoint(const uint8_t* p) {
  const affine a = ld_affine(p);
  return FsPoint{a.x, a.y};
}
// y, z, x (MultiRangeProof/Verifier.hs:31-33) and uChallenge (Verifier.hs:86) of every proof, from the loaded proof
// alone; chal[4][B][32] is what vbatch_begin_async otherwise takes from the host.  Lanes 0..B-1 walk the chain
// y -> z -> x, lanes B..2B-1 hash u.  msg: FS_MSG_WORDS x B words.
__global__ void __launch_bounds__(256) k_fs_initial(VDims d, int style, const uint8_t* __restrict__ sc, const uint8_t* __restrict__ pts,
                                                   uint32_t* __restrict__ msg, uint8_t* __restrict__ chal) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t B = d.B;
  if (t >= 2 * d.B) return;
  if (t < d.B) {
    const uint32_t p = t;
    u256 y, z, x;
    fs_challenges_yzx(style, ld_fspoint(pts + ((size_t)VP_A * B + p) * 64), ld_fspoint(pts + ((size_t)VP_S * B + p) * 64),
                      ld_fspoint(pts + ((size_t)VP_T1 * B + p) * 64), ld_fspoint(pts + ((size_t)VP_T2 * B + p) * 64),
                      msg + p, B, y, z, x);
    st_u256(chal + ((size_t)0 * B + p) * 32, y);
    st_u256(chal + ((size_t)1 * B + p) * 32, z);
    st_u256(chal + ((size_t)2 * B + p) * 32, x);
  } else {
    const uint32_t p = t - d.B;
    const u256 u = fs_challenge_u(style, ld_u256(sc + ((size_t)VS_TAU * B + p) * 32), ld_u256(sc + ((size_t)VS_MU * B + p) * 32),
                                  ld_u256(sc + ((size_t)VS_T * B + p) * 32), msg + (size_t)FS_U_WORD * B + p, B);
    st_u256(chal + ((size_t)3 * B + p) * 32, u);
  }
}

// The openings show _r <> L_r <> R_r of the rounds' messages, hashed ahead of the rounds (side stream):
// pre[(r * FS_PRE_WORDS + w) * B + p].  buf: FS_OPEN_WORDS x B words.
constexpr uint32_t FS_OPEN_WORDS = 96;  // 14 + 2 x 167 bytes of text
__global__ void __launch_bounds__(256) k_fs_openings(VDims d, int style, const uint8_t* __restrict__ pts, uint32_t* __restrict__ buf,
                                                    uint32_t* __restrict__ pre) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.B) return;
  const size_t B = d.B;
#pragma unroll 1
  for (uint32_t r = 0; r < d.k; r++)
    fs_round_opening(style, ld_fspoint(pts + ((size_t)(VP_L0 + r) * B + p) * 64),
                     ld_fspoint(pts + ((size_t)(VP_L0 + d.k + r) * B + p) * 64), buf + p, B,
                     pre + (size_t)r * FS_PRE_WORDS * B + p, B);
}

// x_r = shamirX' acc L_r R_r = oracle (show _r <> L_r <> R_r <> acc)  (mkChallenges, InnerProductProof/Verifier.hs:44-53;
// Utils.hs:167-170): the hashed opening continued with acc, the running commitment the previous kernel left in acc_aff
__global__ void __launch_bounds__(256) k_fs_round(VDims d, uint32_t round, int style, const uint32_t* __restrict__ pre,
                                                 const uint8_t* __restrict__ acc_aff, uint32_t* __restrict__ msg,
                                                 uint8_t* __restrict__ xk /*[B][32]*/) {
  __builtin_amdgcn_s_setprio(3);
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.B) return;
  const size_t B = d.B;
  st_u256(xk + (size_t)p * 32, fs_round_challenge(style, pre + (size_t)round * FS_PRE_WORDS * B + p, B,
                                                  ld_fspoint(acc_aff + (size_t)p * 64), msg + p, B));
}

// Input validation at load time: bad[p] = 1 when proof p holds a point that is neither O nor a canonically encoded
// (x, y < p) point of the curve, or a scalar >= r.  The reference's types cannot represent such values (Fq / Fr are
// always reduced; an off-curve `A x y` would run through its affine formulas, for which no parity can be claimed);
// here such a proof simply does not verify.  One thread per (row, proof); rows 0..4 are the scalar rows.
__global__ void __launch_bounds__(256) k_vcheck(uint32_t B, uint32_t npts, const uint8_t* __restrict__ sc,
                                               const uint8_t* __restrict__ pts, uint8_t* __restrict__ bad) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)(5 + npts) * B) return;
  uint32_t row = (uint32_t)(t / B), p = (uint32_t)(t % B);
  bool ok;
  if (row < 5) {
    ok = !u256_geq(ld_u256(sc + ((size_t)row * B + p) * 32), Fr::modulus());
  } else {
    affine a = ld_affine(pts + ((size_t)(row - 5) * B + p) * 64);
    const u256 q = fq_p();
    ok = affine_is_inf(a) || (!u256_geq(a.x, q) && !u256_geq(a.y, q) && affine_on_curve(a));
  }
  if (!ok) bad[p] = 1;
}

// verdict[p] = (acc1 == acc_k) && (acc2 == O) && the proof's encoding was valid
__global__ void __launch_bounds__(256) k_verdict(VDims d, const xyzz* __restrict__ acc1, const xyzz* __restrict__ acc2,
                                                const uint8_t* __restrict__ acc_aff, const xyzz* __restrict__ acc_x,
                                                const uint8_t* __restrict__ bad, uint8_t* __restrict__ verdict,
                                                uint8_t* __restrict__ host_out /*pinned host copy, or null*/) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.B) return;
  // the running commitment: affine bytes, or (latency path) XYZZ
  xyzz e1 = acc_x ? xyzz_add(ld_xyzz(&acc1[p]), xyzz_neg(ld_xyzz(&acc_x[p])))
                  : xyzz_madd(ld_xyzz(&acc1[p]), affine_neg(ld_affine(acc_aff + (size_t)p * 64)));
  bool ok = xyzz_is_inf(e1) && xyzz_is_inf(ld_xyzz(&acc2[p])) && bad[p] == 0;
  verdict[p] = ok ? 1 : 0;
  if (host_out) host_out[p] = ok ? 1 : 0;
}

}  // namespace bp

using namespace bp;

#define VB_GRID(n) dim3((unsigned)((((size_t)(n)) + 255) / 256)), dim3(256), 0, vb->stream

namespace bp {

// HIP-event timing of k_fixed_multiexp on the launching stream (bp_ctx_profile)
static inline void prof_begin(bp_vbatch* vb, int slot, uint32_t nterms) {
  if (!vb->ctx->profile) return;
  (void)hipEventRecord(vb->ev[slot][0], vb->stream);
  // one mixed add per non-zero digit: 256 / wb windows per term
  vb->ev_madds[slot] = (double)vb->d.B * nterms * (256.0 / vb->d.wb);
}
static inline void prof_end(bp_vbatch* vb, int slot) {
  if (!vb->ctx->profile) return;
  (void)hipEventRecord(vb->ev[slot][1], vb->stream);
  vb->ev_pending[slot] = true;
}

#define VB_GRID_ON(n, strm) dim3((unsigned)((((size_t)(n)) + 255) / 256)), dim3(256), 0, strm
static inline void launch_fixed(bp_vbatch* vb, size_t nthreads, uint32_t nterms, uint32_t lpp, const uint32_t* dig,
                                const uint32_t* tb, xyzz* acc, hipStream_t strm = nullptr) {
  const uint8_t* table = vb->ctx->d_table;
  if (!strm) strm = vb->stream;
  if (vb->split) {  // few proofs: about six mixed additions per lane
    const uint32_t nf = nterms * (uint32_t)(fb_windows((int)vb->d.wb));
    uint32_t nblk = (nf + 256u * 6u - 1) / (256u * 6u);
    nblk = nblk < 1 ? 1 : (nblk > 64 ? 64 : nblk);
    dim3 grid(nblk, vb->d.B), block(256);
    if (vb->d.wb == 20)
      hipLaunchKernelGGL(k_fixed_flat<20>, grid, block, 0, strm, vb->d.B, nterms, dig, tb, table, vb->fpart, acc);
    else if (vb->d.wb == 16)
      hipLaunchKernelGGL(k_fixed_flat<16>, grid, block, 0, strm, vb->d.B, nterms, dig, tb, table, vb->fpart, acc);
    else if (vb->d.wb == 12)
      hipLaunchKernelGGL(k_fixed_flat<12>, grid, block, 0, strm, vb->d.B, nterms, dig, tb, table, vb->fpart, acc);
    else
      hipLaunchKernelGGL(k_fixed_flat<8>, grid, block, 0, strm, vb->d.B, nterms, dig, tb, table, vb->fpart, acc);
    if (nblk > 1) hipLaunchKernelGGL(k_fixed_sum, dim3(vb->d.B), dim3(64), 0, strm, nblk, vb->fpart, acc);
    return;
  }
  if (vb->d.wb == 20) {
    // diagnostic: BP_FIXED_LDS=<bytes> of unused dynamic LDS per workgroup lowers the kernel's occupancy (above 80 KB:
    // one wave per SIMD), to tell a latency-bound gather from an issue-bound loop
    static const int lds_pad = [] {
      const char* e = getenv("BP_FIXED_LDS");
      int v = e ? atoi(e) : 0;
      if (v > 0)
        (void)hipFuncSetAttribute((const void*)k_fixed_multiexp<20>, hipFuncAttributeMaxDynamicSharedMemorySize, v);
      return v;
    }();
    hipLaunchKernelGGL(k_fixed_multiexp<20>, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), (size_t)lds_pad, strm,
                       vb->d.B, nterms, lpp, dig, tb, table, acc);
  } else if (vb->d.wb == 16)
    hipLaunchKernelGGL(k_fixed_multiexp<16>, VB_GRID_ON(nthreads, strm), vb->d.B, nterms, lpp, dig, tb, table, acc);
  else if (vb->d.wb == 12)
    hipLaunchKernelGGL(k_fixed_multiexp<12>, VB_GRID_ON(nthreads, strm), vb->d.B, nterms, lpp, dig, tb, table, acc);
  else
    hipLaunchKernelGGL(k_fixed_multiexp<8>, VB_GRID_ON(nthreads, strm), vb->d.B, nterms, lpp, dig, tb, table, acc);
}

// The t-polynomial check: acc2 = (t - delta) G + tau H - sum z^(j+2) V_j - x T1 - x^2 T2.  Needs k_vprep0's
// constants: the caller launches it on `strm` only once those are complete (same stream, or a later host visit).
static inline void launch_tcheck(bp_vbatch* vb, hipStream_t strm) {
  VDims d = vb->d;
  size_t B = d.B;
  hipLaunchKernelGGL(k_vprep_t, VB_GRID_ON(B, strm), d, vb->sc, vb->cst, vb->zpow, vb->dig3b, vb->vs);
  launch_fixed(vb, B, vb->nt3b, 1u, vb->dig3b, vb->tb3b, vb->acc2, strm);
  if (vb->split) {
    hipLaunchKernelGGL(k_tcheck_bits, dim3(d.B), dim3(256), 0, strm, d, vb->vs, vb->sh, vb->acc2);
    return;
  }
  const uint32_t nt = vt_tables(d);
  for (uint32_t t = 0; t < nt; t += 2) {
    if (t + 1 < nt)
      hipLaunchKernelGGL(k_var_terms<2>, VB_GRID_ON(B, strm), d, t, vb->vs, vb->rtab, vb->acc2);
    else
      hipLaunchKernelGGL(k_var_terms<1>, VB_GRID_ON(B, strm), d, t, vb->vs, vb->rtab, vb->acc2);
  }
}

int vbatch_load_async(bp_vbatch* vb, const uint8_t* scalars, const uint8_t* points) {
  bp_ctx* ctx = vb->ctx;
  size_t B = vb->d.B;
  if (vb->epoch != ctx->gen_epoch) {
    ctx->err = "the context's generators were replaced after this batch object was created: create a new one";
    return BP_ERR_STATE;
  }
  if (vb->tab_pending) {  // tables of the previous proofs
    BP_TRY(hipStreamSynchronize(vb->aux));
    vb->tab_pending = false;
  }
  BP_TRY(hipMemcpyAsync(vb->sc, scalars, 5 * B * 32, hipMemcpyHostToDevice, vb->stream));
  BP_TRY(hipMemcpyAsync(vb->pts, points, (size_t)vb->npts * B * 64, hipMemcpyHostToDevice, vb->stream));
  BP_TRY(hipMemsetAsync(vb->bad, 0, B, vb->stream));
  hipLaunchKernelGGL(k_vcheck, VB_GRID((size_t)(5 + vb->npts) * B), (uint32_t)B, vb->npts, vb->sc, vb->pts, vb->bad);
  BP_TRY(hipGetLastError());
  vb->stage = 1;
  return BP_OK;
}

// Work of a pass that needs the proof's points only, on the side stream: the decimal text of L_r, R_r for the host's
// round transcripts (lr_text: pinned [2k][B][2][80], or null) and the rounds' joint tables.  The batch job calls this
// for every chunk when a pass starts, so the GPU is busy while the host hashes the first challenges; otherwise
// vbatch_begin_async does.  Joined by the host before round 0.
int vbatch_prepare_async(bp_vbatch* vb, uint8_t* lr_text) {
  bp_ctx* ctx = vb->ctx;
  if (vb->stage < 1) {
    ctx->err = "bp_vbatch prepare before bp_vbatch_load";
    return BP_ERR_STATE;
  }
  VDims d = vb->d;
  size_t B = d.B;
  if (vb->tab_pending) return BP_OK;
  if (vb->t_pending) {  // a previous pass's side work
    BP_TRY(hipStreamSynchronize(vb->aux));
    vb->t_pending = false;
  }
  if (vb->split) {
    // every row from S on; the host formats L_r, R_r itself on this path (no text kernel)
    const uint32_t cnt = (vb->npts - VP_S) * d.B;
    int rc = ipp_bit_tables(ctx, vb->pts + (size_t)VP_S * B * 64, cnt, vb->sh, vb->aux);
    if (rc != BP_OK) return rc;
  } else {
    if (lr_text && vb->direct_host && d.k) {
      size_t cnt = (size_t)2 * d.k * B * 2;
      hipLaunchKernelGGL(k_dec_numbers, VB_GRID_ON(cnt, vb->aux), vb->pts + (size_t)VP_L0 * B * 64, (uint32_t)cnt, lr_text);
    }
    // every joint table but S's, which stage 1 needs first and builds on its own stream
    hipLaunchKernelGGL(k_round_tables, VB_GRID_ON(B, vb->aux), d, vb->pts, vb->rtx, vb->rpre, vb->rtab, 0u, joint_tables(d), d.k);
  }
  BP_TRY(hipGetLastError());
  vb->tab_pending = true;
  return BP_OK;
}

// The decimal text of what y, z, x, u hash from the proof itself - A, S, T1, T2 (rows 0..3: records [row][p][2]) and
// tBlinding, mu, t (records [8 B + row][p]) - into pinned host memory, on the batch's own stream (nothing else is
// queued there when a pass starts, so the host gets the text after a fraction of a millisecond).
int vbatch_init_text_async(bp_vbatch* vb, uint8_t* init_text) {
  bp_ctx* ctx = vb->ctx;
  if (vb->stage < 1 || !vb->direct_host) {
    ctx->err = "bp_vbatch init text: needs a loaded batch with pinned outputs";
    return BP_ERR_STATE;
  }
  const size_t B = vb->d.B;
  hipLaunchKernelGGL(k_dec_numbers, VB_GRID(8 * B), vb->pts, (uint32_t)(8 * B), init_text);
  hipLaunchKernelGGL(k_dec_numbers, VB_GRID(3 * B), vb->sc, (uint32_t)(3 * B), init_text + 8 * B * DEC_REC);
  BP_TRY(hipGetLastError());
  return BP_OK;
}

int vbatch_begin_async(bp_vbatch* vb, const uint8_t* challenges, uint8_t* out_commitment, uint8_t* out_text,
                       uint8_t* lr_text) {
  bp_ctx* ctx = vb->ctx;
  if (vb->stage < 1) {
    ctx->err = "bp_vbatch_begin before bp_vbatch_load";
    return BP_ERR_STATE;
  }
  if (vb->epoch != ctx->gen_epoch) {
    ctx->err = "the context's generators were replaced after this batch object was created: create a new one";
    return BP_ERR_STATE;
  }
  VDims d = vb->d;
  size_t B = d.B;
  if (vb->t_pending) {  // begun again before the previous pass finished: its side work still reads the constants
    BP_TRY(hipStreamSynchronize(vb->aux));
    vb->t_pending = false;
  }
  {
    int rc = vbatch_prepare_async(vb, lr_text);  // no-op when the caller has prepared this pass already
    if (rc != BP_OK) return rc;
  }
  // pinned caller buffers are read by the kernel itself (one 32-byte read per proof and challenge over the
  // host link): no copy command enters the queue
  const uint8_t* chal = challenges;
  if (!vb->direct_host) {
    BP_TRY(hipMemcpyAsync(vb->chal, challenges, 4 * B * 32, hipMemcpyHostToDevice, vb->stream));
    chal = vb->chal;
  }
  uint32_t lpp = vbatch_lpp(d.B);
  uint8_t* direct = vb->direct_host ? out_commitment : nullptr;
  if (vb->split) {
    // the two inversions of this stage on the host (`challenges` is host memory in both modes)
    vbatch_host_for(vb, B, [&](size_t p0, size_t p1) {
      for (size_t p = p0; p < p1; p++) {
        u256 y;
        memcpy(&y, challenges + ((size_t)0 * B + p) * 32, 32);
        y = Fr::to_mont(Fr::reduce_once(y));
        u256 yi = host::fr_inv_mont_host(y), ym1i = host::fr_inv_mont_host(Fr::sub(y, Fr::one()));
        memcpy(vb->hinv + ((size_t)HI_YINV * B + p) * 32, &yi, 32);
        memcpy(vb->hinv + ((size_t)HI_YM1INV * B + p) * 32, &ym1i, 32);
      }
    });
    hipLaunchKernelGGL(k_vprep0_split, dim3((d.B + 63) / 64), dim3(64), 0, vb->stream, d, chal, vb->hinv, vb->cst, vb->zpow);
    hipLaunchKernelGGL(k_vprep0_pow, VB_GRID((size_t)d.nm * B), d, vb->cst, vb->ypowinv);
    hipLaunchKernelGGL(k_vprep1, VB_GRID((size_t)d.nm * B), d, vb->sc, vb->cst, vb->ypowinv, vb->zpow, vb->dig1);
    launch_fixed(vb, B * lpp, vb->nt1, lpp, vb->dig1, vb->tb1, vb->acc1);
    // the bit tables come from the side stream (launched at prepare, ~0.5 ms of doublings): host-side join, which
    // also covers the rounds' and the t-check's needs
    if (vb->tab_pending && vb->aux != vb->stream) BP_TRY(hipStreamSynchronize(vb->aux));
    vb->tab_pending = false;
    hipLaunchKernelGGL(k_stage1_tail_bits, dim3(d.B), dim3(128), 0, vb->stream, d, vb->pts, vb->cst, vb->acc1, vb->sh,
                       vb->acc_x, vb->hx);
    BP_TRY(hipGetLastError());
    vb->collect_pending = true;
    vb->stage = 2;
    vb->begun_complete = false;
    return BP_OK;
  }
  static const int pair_env = getenv("BP_ROUND_PAIR") ? atoi(getenv("BP_ROUND_PAIR")) : -1;
  const bool tail_pair = pair_env < 0 ? B <= 8192 : pair_env != 0;
  // S's joint table (one composite entry), in front of the stage that uses it on the same stream
  if (!tail_pair)
    hipLaunchKernelGGL(k_round_tables, VB_GRID(B), d, vb->pts, vb->rtx, vb->rpre, vb->rtab, d.k, d.k + 1, 0xFFFFFFFFu);
  hipLaunchKernelGGL(k_vprep0, VB_GRID(B), d, chal, vb->cst, vb->ypowinv, vb->zpow, true);
  hipLaunchKernelGGL(k_vprep1, VB_GRID((size_t)d.nm * B), d, vb->sc, vb->cst, vb->ypowinv, vb->zpow, vb->dig1);
  prof_begin(vb, 0, vb->nt1);
  launch_fixed(vb, B * lpp, vb->nt1, lpp, vb->dig1, vb->tb1, vb->acc1);
  prof_end(vb, 0);
  // A device-to-host copy command after the kernel would serialise the chunks' streams against each
  // other (measured: benchmarks/concurrency_test3.hip); with pinned outputs the kernel stores its
  // result to host memory itself.
  // chunks of at most 8,192 proofs: the chain of x*S on pairs of lanes, as in the rounds (BP_ROUND_PAIR)
  if (tail_pair)
    hipLaunchKernelGGL(k_stage1_tail_pair, dim3((unsigned)((B + RP_PROOFS - 1) / RP_PROOFS)), dim3(256), 0, vb->stream, d, vb->pts,
                       vb->cst, vb->acc1, (uint8_t*)vb->vtab, vb->acc_aff, direct, vb->direct_host ? out_text : nullptr);
  else {
    hipLaunchKernelGGL(k_stage1_tail, VB_GRID(B), d, vb->pts, vb->cst, vb->acc1, vb->rtab, vb->acc_aff, direct,
                       vb->direct_host ? out_text : nullptr);
  }
  BP_TRY(hipGetLastError());
  if (!direct) BP_TRY(hipMemcpyAsync(out_commitment, vb->acc_aff, B * 64, hipMemcpyDeviceToHost, vb->stream));
  vb->stage = 2;
  vb->begun_complete = false;  // set by vbatch_wait
  return BP_OK;
}

int vbatch_round_async(bp_vbatch* vb, uint32_t round, const uint8_t* x, uint8_t* out_commitment, uint8_t* out_text) {
  bp_ctx* ctx = vb->ctx;
  if (vb->stage != (int)(2 + round) || round >= vb->d.k) {
    ctx->err = "bp_vbatch_round out of order";
    return BP_ERR_STATE;
  }
  VDims d = vb->d;
  size_t B = d.B;
  if (vb->tab_pending) {
    // launched beside stage 1, several milliseconds ago: a host-side join keeps event/barrier packets
    // out of the queues (on a shared stream the order alone would do)
    if (vb->aux != vb->stream) BP_TRY(hipStreamSynchronize(vb->aux));
    vb->tab_pending = false;
  }
  if (round == 0 && vb->begun_complete && !vb->t_pending && vb->aux != vb->stream) {
    // the stream has drained since begin (the caller saw its output), so k_vprep0's constants are in place
    launch_tcheck(vb, vb->aux);
    vb->t_pending = true;
  }
  const uint8_t* xin = x;
  if (!vb->direct_host) {
    BP_TRY(hipMemcpyAsync(vb->xk, x, B * 32, hipMemcpyHostToDevice, vb->stream));
    xin = vb->xk;
  }
  uint8_t* direct = vb->direct_host ? out_commitment : nullptr;
  // The two inversions of a round are per lane: the kernel is a latency chain the host waits for, and sharing
  // them across the workgroup (batchinv.h) lengthens it (1.58 -> 1.71 ms).  BP_ROUND_BATCHINV=1 selects the shared form.
  static const bool round_batch = getenv("BP_ROUND_BATCHINV") && atoi(getenv("BP_ROUND_BATCHINV")) != 0;
  if (vb->split) {
    vbatch_host_for(vb, B, [&](size_t p0, size_t p1) {  // x_r^-1 on the host (`x` is host memory in both modes)
      for (size_t p = p0; p < p1; p++) {
        u256 xv;
        memcpy(&xv, x + p * 32, 32);
        xv = host::fr_inv_mont_host(Fr::to_mont(Fr::reduce_once(xv)));
        memcpy(vb->hinv + ((size_t)HI_XINV * B + p) * 32, &xv, 32);
      }
    });
    hipLaunchKernelGGL(k_round_bits, dim3(d.B), dim3(256), 0, vb->stream, d, round, xin, vb->hinv, vb->xk_m, vb->xkinv_m,
                       vb->sh, vb->acc_x, vb->hx);
    BP_TRY(hipGetLastError());
    vb->collect_pending = true;
    vb->stage++;
    return BP_OK;
  }
  // Two lanes per proof (k_round_pair) shorten the chain the host waits for but issue about a third more instructions:
  // a gain while the GPU has idle issue slots (1,024 proofs 8.7 -> 7.4 ms per pass, 8,192 11.2 -> 9.9, 32,768 in four
  // chunks 16.6 -> 15.9), a loss once the chunks of a large batch fill it (65,536 in four chunks 25.1 -> 26.1 ms).
  // BP_ROUND_PAIR=0 / 1 forces one form.
  static const int pair_env = getenv("BP_ROUND_PAIR") ? atoi(getenv("BP_ROUND_PAIR")) : -1;
  const bool round_pair = pair_env < 0 ? B <= 8192 : pair_env != 0;
  if (round_pair && !round_batch)
    hipLaunchKernelGGL(k_round_pair, dim3((unsigned)((B + RP_PROOFS - 1) / RP_PROOFS)), dim3(256), 0, vb->stream, d, round,
                       xin, vb->xk_m, vb->xkinv_m, vb->rtab, vb->acc_aff, direct, vb->direct_host ? out_text : nullptr);
  else if (round_batch)
    hipLaunchKernelGGL(k_round<true>, VB_GRID(B), d, round, xin, vb->xk_m, vb->xkinv_m, vb->rtab, vb->acc_aff, direct,
                       vb->direct_host ? out_text : nullptr);
  else
    hipLaunchKernelGGL(k_round<false>, VB_GRID(B), d, round, xin, vb->xk_m, vb->xkinv_m, vb->rtab, vb->acc_aff, direct,
                       vb->direct_host ? out_text : nullptr);
  BP_TRY(hipGetLastError());
  if (!direct) BP_TRY(hipMemcpyAsync(out_commitment, vb->acc_aff, B * 64, hipMemcpyDeviceToHost, vb->stream));
  vb->stage++;
  return BP_OK;
}

int vbatch_finish_async(bp_vbatch* vb, uint8_t* verdicts) {
  bp_ctx* ctx = vb->ctx;
  if (vb->stage != (int)(2 + vb->d.k)) {
    ctx->err = "bp_vbatch_finish before all rounds";
    return BP_ERR_STATE;
  }
  VDims d = vb->d;
  size_t B = d.B;
  hipLaunchKernelGGL(k_vprep3a, VB_GRID(B), d, vb->xk_m, vb->xkinv_m, vb->cst, vb->stab);
  hipLaunchKernelGGL(k_vprep3, VB_GRID((size_t)d.nm * B), d, vb->sc, vb->cst, vb->ypowinv, vb->stab, vb->dig3a);
  uint32_t lpp = vbatch_lpp(d.B);
  prof_begin(vb, 1, vb->nt3a);
  launch_fixed(vb, B * lpp, vb->nt3a, lpp, vb->dig3a, vb->tb3a, vb->acc1);
  prof_end(vb, 1);
  if (vb->t_pending) {
    // launched on the side stream at round 0, many milliseconds ago: this wait is a formality, and it keeps
    // event/barrier packets (which break the overlap of the chunk streams) out of the queues
    BP_TRY(hipStreamSynchronize(vb->aux));
    vb->t_pending = false;
  } else {
    launch_tcheck(vb, vb->stream);
  }
  uint8_t* direct = vb->direct_host ? verdicts : nullptr;
  hipLaunchKernelGGL(k_verdict, VB_GRID(B), d, vb->acc1, vb->acc2, vb->acc_aff, vb->split ? vb->acc_x : nullptr, vb->bad,
                     vb->verdict, direct);
  BP_TRY(hipGetLastError());
  if (!direct) BP_TRY(hipMemcpyAsync(verdicts, vb->verdict, B, hipMemcpyDeviceToHost, vb->stream));
  vb->stage = 1;  // may be begun again with the same loaded proofs
  return BP_OK;
}

// A whole pass with the transcript on the device: every kernel of the pass goes onto the batch's stream in order, the
// challenges pass from kernel to kernel in HBM and the host only collects the verdicts (vbatch_wait, then `verdicts`
// holds them: pinned memory the last kernel stores to, or the target of a copy on the same stream).
int vbatch_run_fsdev_async(bp_vbatch* vb, uint8_t* verdicts) {
  bp_ctx* ctx = vb->ctx;
  if (vb->stage < 1 || vb->split) {
    ctx->err = vb->split ? "device transcript: not for the latency path" : "device transcript before bp_vbatch_load";
    return BP_ERR_STATE;
  }
  if (vb->epoch != ctx->gen_epoch) {
    ctx->err = "the context's generators were replaced after this batch object was created: create a new one";
    return BP_ERR_STATE;
  }
  if (vb->t_pending || vb->tab_pending) {  // side work of an earlier pass of the host-transcript form
    BP_TRY(hipStreamSynchronize(vb->aux));
    vb->t_pending = vb->tab_pending = false;
  }
  const VDims d = vb->d;
  const size_t B = d.B;
  const int style = ctx->show_style;
  hipStream_t strm = vb->stream;
  // Two lanes per proof in the chains (k_round_pair, k_stage1_tail_pair: a third more instructions for a shorter chain)
  // while the job's single chunk leaves SIMDs without a wave: up to 32,768 proofs (16,384: 9.7 -> 8.4 ms, 32,768: 12.6
  // -> 11.4 ms; 65,536: 17.5 -> 21.2 ms, so not there).  The host form, with four chunks in flight, stops at 8,192.
  static const int pair_env = getenv("BP_ROUND_PAIR") ? atoi(getenv("BP_ROUND_PAIR")) : -1;
  const bool pair = pair_env < 0 ? B <= 32768 : pair_env != 0;
  // The side stream carries what the main chain does not wait for: every joint table but S's (wanted by round 0 and by
  // the t-check) and the t-polynomial check (wants y, z, x: k_vprep0's constants).  Its chains run beside the main
  // stream's, a second wave on the SIMDs.  BP_FS_SIDE=0 keeps everything on one stream.
  static const bool side_env = !(getenv("BP_FS_SIDE") && atoi(getenv("BP_FS_SIDE")) == 0);
  static const bool fs_round_batchinv = getenv("BP_ROUND_BATCHINV") && atoi(getenv("BP_ROUND_BATCHINV")) != 0;
  const bool side = side_env && vb->aux != strm;
  hipStream_t sd = side ? vb->aux : strm;
  if (d.k) hipLaunchKernelGGL(k_fs_openings, VB_GRID_ON(B, sd), d, style, vb->pts, vb->fsmsg2, vb->fspre);
  hipLaunchKernelGGL(k_round_tables, VB_GRID_ON(B, sd), d, vb->pts, vb->rtx, vb->rpre, vb->rtab, 0u, joint_tables(d), d.k);
  if (side) BP_TRY(hipEventRecord(vb->fs_ev[1], sd));
  hipLaunchKernelGGL(k_fs_initial, VB_GRID(2 * B), d, style, vb->sc, vb->pts, vb->fsmsg, vb->chal);
  hipLaunchKernelGGL(k_round_tables, VB_GRID(B), d, vb->pts, vb->rtx, vb->rpre, vb->rtab, d.k, d.k + 1, 0xFFFFFFFFu);
  hipLaunchKernelGGL(k_vprep0, VB_GRID(B), d, vb->chal, vb->cst, vb->ypowinv, vb->zpow, true);
  if (side) {
    BP_TRY(hipEventRecord(vb->fs_ev[0], strm));
    BP_TRY(hipStreamWaitEvent(sd, vb->fs_ev[0], 0));
  }
  launch_tcheck(vb, sd);
  if (side) BP_TRY(hipEventRecord(vb->fs_ev[2], sd));
  hipLaunchKernelGGL(k_vprep1, VB_GRID((size_t)d.nm * B), d, vb->sc, vb->cst, vb->ypowinv, vb->zpow, vb->dig1);
  const uint32_t lpp = vbatch_lpp(d.B);
  prof_begin(vb, 0, vb->nt1);
  launch_fixed(vb, B * lpp, vb->nt1, lpp, vb->dig1, vb->tb1, vb->acc1);
  prof_end(vb, 0);
  if (pair)
    hipLaunchKernelGGL(k_stage1_tail_pair, dim3((unsigned)((B + RP_PROOFS - 1) / RP_PROOFS)), dim3(256), 0, strm, d, vb->pts,
                       vb->cst, vb->acc1, (uint8_t*)vb->vtab, vb->acc_aff, (uint8_t*)nullptr, (uint8_t*)nullptr);
  else
    hipLaunchKernelGGL(k_stage1_tail, VB_GRID(B), d, vb->pts, vb->cst, vb->acc1, vb->rtab, vb->acc_aff, (uint8_t*)nullptr,
                       (uint8_t*)nullptr);
  // the rounds read the side stream's tables
  if (side) BP_TRY(hipStreamWaitEvent(strm, vb->fs_ev[1], 0));
  for (uint32_t r = 0; r < d.k; r++) {
    hipLaunchKernelGGL(k_fs_round, VB_GRID(B), d, r, style, vb->fspre, vb->acc_aff, vb->fsmsg, vb->xk);
    if (pair)
      hipLaunchKernelGGL(k_round_pair, dim3((unsigned)((B + RP_PROOFS - 1) / RP_PROOFS)), dim3(256), 0, strm, d, r, vb->xk,
                         vb->xk_m, vb->xkinv_m, vb->rtab, vb->acc_aff, (uint8_t*)nullptr, (uint8_t*)nullptr);
    else if (fs_round_batchinv)
      hipLaunchKernelGGL(k_round<true>, VB_GRID(B), d, r, vb->xk, vb->xk_m, vb->xkinv_m, vb->rtab, vb->acc_aff,
                         (uint8_t*)nullptr, (uint8_t*)nullptr);
    else
      hipLaunchKernelGGL(k_round<false>, VB_GRID(B), d, r, vb->xk, vb->xk_m, vb->xkinv_m, vb->rtab, vb->acc_aff,
                         (uint8_t*)nullptr, (uint8_t*)nullptr);
  }
  hipLaunchKernelGGL(k_vprep3a, VB_GRID(B), d, vb->xk_m, vb->xkinv_m, vb->cst, vb->stab);
  hipLaunchKernelGGL(k_vprep3, VB_GRID((size_t)d.nm * B), d, vb->sc, vb->cst, vb->ypowinv, vb->stab, vb->dig3a);
  prof_begin(vb, 1, vb->nt3a);
  launch_fixed(vb, B * lpp, vb->nt3a, lpp, vb->dig3a, vb->tb3a, vb->acc1);
  prof_end(vb, 1);
  if (side) BP_TRY(hipStreamWaitEvent(strm, vb->fs_ev[2], 0));  // acc2 of the t-check
  uint8_t* direct = vb->direct_host ? verdicts : nullptr;
  hipLaunchKernelGGL(k_verdict, VB_GRID(B), d, vb->acc1, vb->acc2, vb->acc_aff, (const xyzz*)nullptr, vb->bad, vb->verdict,
                     direct);
  BP_TRY(hipGetLastError());
  if (!direct) BP_TRY(hipMemcpyAsync(verdicts, vb->verdict, B, hipMemcpyDeviceToHost, strm));
  vb->stage = 1;
  vb->begun_complete = false;
  return BP_OK;
}

// Host-side join of the pass's side work (vbatch_prepare_async: the decimal text of L_r, R_r and the rounds' joint
// tables, on vb->aux).  The caller must do this BEFORE it reads lr_text for the round-0 transcript: vbatch_wait only
// covers vb->stream, and the text is written on the side stream.
int vbatch_join_side(bp_vbatch* vb) {
  bp_ctx* ctx = vb->ctx;
  if (vb->tab_pending) {
    if (vb->aux != vb->stream) BP_TRY(hipStreamSynchronize(vb->aux));
    vb->tab_pending = false;
  }
  return BP_OK;
}

int vbatch_collect(bp_vbatch* vb, uint8_t* out_commitment) {
  if (!vb->split || !vb->collect_pending) return BP_OK;
  vbatch_host_for(vb, vb->d.B, [&](size_t p0, size_t p1) {
    for (size_t p = p0; p < p1; p++) {
      affine a = host::xyzz_to_affine_host(vb->hx[p]);
      memcpy(out_commitment + p * 64, &a, 64);
    }
  });
  vb->collect_pending = false;
  return BP_OK;
}

// A stream query, not an event: recording an event behind the kernels puts a marker packet into the queue,
// and such packets were measured to break the overlap between chunk streams just like copy commands do.
bool vbatch_ready(bp_vbatch* vb) { return hipStreamQuery(vb->stream) == hipSuccess; }

int vbatch_wait(bp_vbatch* vb) {
  bp_ctx* ctx = vb->ctx;
  BP_TRY(hipStreamSynchronize(vb->stream));
  if (vb->stage == 2) vb->begun_complete = true;
  for (int s = 0; s < 3; s++)
    if (vb->ev_pending[s]) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, vb->ev[s][0], vb->ev[s][1]) == hipSuccess) {
        ctx->prof_ms[0] += ms;
        ctx->prof_calls[0] += 1;
        ctx->prof_madds[0] += vb->ev_madds[s];
      }
      vb->ev_pending[s] = false;
    }
  return BP_OK;
}

}  // namespace bp

extern "C" {

int bp_vbatch_create(bp_ctx* ctx, uint32_t n_bits, uint32_t m, uint32_t batch, bp_vbatch** out) {
  return vbatch_create_ex(ctx, n_bits, m, batch, false, out);
}

}  // extern "C"

// pooled_host: the caller (bp_vjob) lends its hashing threads to the latency path's host loops (bp_vbatch::pfor), which
// moves the size at which that path stops paying from ~100 to ~1,000 proofs
int bp::vbatch_create_ex(bp_ctx* ctx, uint32_t n_bits, uint32_t m, uint32_t batch, bool pooled_host, bp_vbatch** out) {
  if (!ctx) return BP_ERR_ARG;
  BP_REQUIRE(out, "null pointer");
  *out = nullptr;
  const uint32_t split_max = pooled_host ? vsplit_max_b() : (vsplit_max_b() < VSPLIT_SERIAL_MAX_B ? vsplit_max_b() : VSPLIT_SERIAL_MAX_B);
  BP_REQUIRE(n_bits >= 1 && n_bits <= 128 && (n_bits & (n_bits - 1)) == 0, "n_bits must be a power of two <= 128");
  BP_REQUIRE(m >= 1 && (m & (m - 1)) == 0, "m must be a power of two (pad commitments with O, Verifier.hs:37-38)");
  BP_REQUIRE(batch >= 1 && batch <= (1u << 22), "batch must be in 1..2^22");
  uint32_t nm = n_bits * m;
  if (ctx->n_gens == 0) {
    ctx->err = "generators not set (bp_ctx_set_generators)";
    return BP_ERR_STATE;
  }
  BP_REQUIRE(nm <= ctx->n_gens, "n_bits * m exceeds the generator count of the context");
  BP_TRY(hipSetDevice(ctx->device));
  bp_vbatch* vb = new bp_vbatch();
  vb->ctx = ctx;
  vb->stream = nullptr;
  VDims& d = vb->d;
  d.B = batch;
  d.n = n_bits;
  d.m = m;
  d.nm = nm;
  d.k = 0;
  while ((1u << d.k) < nm) d.k++;
  d.wb = (uint32_t)ctx->fb_wbits;
  const size_t DW = fb_dwords(ctx->fb_wbits);
  vb->npts = VP_L0 + 2 * d.k + m;
  vb->nt1 = nm + 3;
  vb->nt3a = 2 * nm + 1;
  vb->nt3b = 2;
  size_t B = batch;
  auto carve = [&](void* base) {
    Carver c(base);
    vb->sc = c.take<uint8_t>(5 * B * 32);
    vb->pts = c.take<uint8_t>((size_t)vb->npts * B * 64);
    vb->chal = c.take<uint8_t>(4 * B * 32);
    vb->xk = c.take<uint8_t>(B * 32);
    vb->acc_aff = c.take<uint8_t>(B * 64);
    vb->vs = c.take<uint8_t>((size_t)(m + 2) * B * 32);
    vb->verdict = c.take<uint8_t>(B);
    vb->bad = c.take<uint8_t>(B);
    vb->cst = c.take<u256>((size_t)C_COUNT * B);
    vb->ypowinv = c.take<u256>((size_t)nm * B);
    vb->zpow = c.take<u256>((size_t)(m + 3) * B);
    vb->xk_m = c.take<u256>((size_t)(d.k ? d.k : 1) * B);
    vb->xkinv_m = c.take<u256>((size_t)(d.k ? d.k : 1) * B);
    vb->stab = c.take<u256>(((size_t)1 << (d.k >> 1)) * B + ((size_t)1 << (d.k - (d.k >> 1))) * B);
    vb->dig1 = c.take<uint32_t>((size_t)vb->nt1 * DW * B);
    vb->dig3a = c.take<uint32_t>((size_t)vb->nt3a * DW * B);
    vb->dig3b = c.take<uint32_t>((size_t)vb->nt3b * DW * B);
    vb->tb1 = c.take<uint32_t>(vb->nt1);
    vb->tb3a = c.take<uint32_t>(vb->nt3a);
    vb->tb3b = c.take<uint32_t>(vb->nt3b);
    vb->acc1 = c.take<xyzz>(B);
    vb->acc2 = c.take<xyzz>(B);
    vb->vtab = c.take<xyzz>(B);  // S, phi(S) of k_stage1_tail_pair (128 bytes per proof)
    vb->fsmsg = c.take<uint32_t>((size_t)FS_MSG_WORDS * B);  // the device transcript's message buffers
    vb->fsmsg2 = c.take<uint32_t>((size_t)FS_OPEN_WORDS * B);  // ... of the rounds' openings (side stream)
    vb->fspre = c.take<uint32_t>((size_t)(d.k ? d.k : 1) * FS_PRE_WORDS * B);  // the hashed openings
    vb->rtx = c.take<uint32_t>((size_t)11 * joint_tables(d) * 36 * B);
    vb->rpre = c.take<uint32_t>((size_t)11 * joint_tables(d) * 9 * B);
    vb->rtab = c.take<uint8_t>((size_t)15 * joint_tables(d) * B * 64);
    {
      const size_t sb = B <= split_max ? B : 1;
      vb->sh = c.take<xyzz>((size_t)(vb->npts - VP_S) * sb * GLV_LANE_BITS);  // bit tables of rows S.. (16 KB a point)
      vb->acc_x = c.take<xyzz>(sb);
      vb->fpart = c.take<xyzz>(sb * 64);
    }
    return c.off;
  };
  size_t need = carve(nullptr);
  vb->stream = ctx_chunk_stream(ctx);  // from the context's concurrent set; not owned
  vb->aux = ctx_aux_stream(ctx);
  vb->t_pending = false;
  vb->tab_pending = false;
  vb->begun_complete = false;
  vb->split = batch <= split_max && d.k > 0 && !(getenv("BP_NO_SPLIT") && atoi(getenv("BP_NO_SPLIT")) != 0);
  hipError_t e = hipSuccess;
  vb->done = nullptr;
  e = hipEventCreateWithFlags(&vb->done, hipEventDisableTiming);
  for (int s2 = 0; s2 < 3; s2++) {
    vb->ev_pending[s2] = false;
    vb->ev[s2][0] = vb->ev[s2][1] = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&vb->ev[s2][0]);
    if (e == hipSuccess) e = hipEventCreate(&vb->ev[s2][1]);
  }
  for (int s2 = 0; s2 < 3; s2++) {
    vb->fs_ev[s2] = nullptr;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&vb->fs_ev[s2], hipEventDisableTiming);
  }
  vb->hx = nullptr;
  vb->hinv = nullptr;
  vb->collect_pending = false;
  if (e == hipSuccess && vb->split) {  // pinned: [B] xyzz, then the host's inverses [3][B][32]
    void* hp = nullptr;
    e = hipHostMalloc(&hp, B * sizeof(xyzz) + 3 * B * 32, hipHostMallocDefault);
    if (e == hipSuccess) {
      vb->hx = (xyzz*)hp;
      vb->hinv = (uint8_t*)hp + B * sizeof(xyzz);
    }
  }
  if (e != hipSuccess || vb->buf.reserve(need) != 0) {
    if (vb->hx) (void)hipHostFree(vb->hx);
    delete vb;
    ctx->err = "verifier workspace allocation failed";
    return BP_ERR_NOMEM;
  }
  carve(vb->buf.p);
  // term -> base maps
  uint32_t ng = ctx->n_gens;
  std::vector<uint32_t> tb1(vb->nt1), tb3a(vb->nt3a), tb3b(2);
  for (uint32_t i = 0; i < nm; i++) tb1[i] = fb_base_h(ng, i);
  tb1[nm] = fb_base_gsum(ng, d.k);
  tb1[nm + 1] = fb_base_H(ng);
  tb1[nm + 2] = fb_base_G(ng);
  for (uint32_t i = 0; i < nm; i++) {
    tb3a[i] = fb_base_g(i);
    tb3a[nm + i] = fb_base_h(ng, i);
  }
  tb3a[2 * nm] = fb_base_G(ng);
  tb3b[0] = fb_base_G(ng);
  tb3b[1] = fb_base_H(ng);
  e = hipMemcpyAsync(vb->tb1, tb1.data(), tb1.size() * 4, hipMemcpyHostToDevice, vb->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(vb->tb3a, tb3a.data(), tb3a.size() * 4, hipMemcpyHostToDevice, vb->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(vb->tb3b, tb3b.data(), tb3b.size() * 4, hipMemcpyHostToDevice, vb->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(vb->stream);
  if (e != hipSuccess) {
    vb->buf.release();
    if (vb->hx) (void)hipHostFree(vb->hx);
    delete vb;
    ctx->err = std::string("bp_vbatch_create: ") + hipGetErrorString(e);
    return BP_ERR_DEVICE;
  }
  vb->stage = 0;
  vb->epoch = ctx->gen_epoch;
  vb->direct_host = false;
  *out = vb;
  return BP_OK;
}

extern "C" {

void bp_vbatch_destroy(bp_vbatch* vb) {
  if (!vb) return;
  (void)hipSetDevice(vb->ctx->device);
  (void)hipStreamSynchronize(vb->stream);
  if (vb->t_pending || vb->tab_pending) (void)hipStreamSynchronize(vb->aux);
  vb->buf.release();
  if (vb->hx) (void)hipHostFree(vb->hx);
  for (int s = 0; s < 3; s++)
    for (int j = 0; j < 2; j++)
      if (vb->ev[s][j]) (void)hipEventDestroy(vb->ev[s][j]);
  for (int s = 0; s < 3; s++)
    if (vb->fs_ev[s]) (void)hipEventDestroy(vb->fs_ev[s]);
  if (vb->done) (void)hipEventDestroy(vb->done);
  delete vb;
}

uint32_t bp_vbatch_rounds(const bp_vbatch* vb) { return vb ? vb->d.k : 0; }
uint32_t bp_vbatch_point_rows(const bp_vbatch* vb) { return vb ? vb->npts : 0; }

int bp_vbatch_load(bp_vbatch* vb, const uint8_t* scalars, const uint8_t* points) {
  if (!vb) return BP_ERR_ARG;
  bp_ctx* ctx = vb->ctx;
  BP_REQUIRE(scalars && points, "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  int rc = vbatch_load_async(vb, scalars, points);
  return rc != BP_OK ? rc : vbatch_wait(vb);
}

int bp_vbatch_begin(bp_vbatch* vb, const uint8_t* challenges, uint8_t* out_commitment) {
  if (!vb) return BP_ERR_ARG;
  bp_ctx* ctx = vb->ctx;
  BP_REQUIRE(challenges && out_commitment, "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  int rc = vbatch_begin_async(vb, challenges, out_commitment, nullptr, nullptr);
  if (rc == BP_OK) rc = vbatch_wait(vb);
  return rc != BP_OK ? rc : vbatch_collect(vb, out_commitment);
}

int bp_vbatch_round(bp_vbatch* vb, uint32_t round, const uint8_t* x, uint8_t* out_commitment) {
  if (!vb) return BP_ERR_ARG;
  bp_ctx* ctx = vb->ctx;
  BP_REQUIRE(x && out_commitment, "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  int rc = vbatch_round_async(vb, round, x, out_commitment, nullptr);
  if (rc == BP_OK) rc = vbatch_wait(vb);
  return rc != BP_OK ? rc : vbatch_collect(vb, out_commitment);
}

int bp_vbatch_finish(bp_vbatch* vb, uint8_t* verdicts) {
  if (!vb) return BP_ERR_ARG;
  bp_ctx* ctx = vb->ctx;
  BP_REQUIRE(verdicts, "null pointer");
  BP_TRY(hipSetDevice(ctx->device));
  int rc = vbatch_finish_async(vb, verdicts);
  return rc != BP_OK ? rc : vbatch_wait(vb);
}

}  // extern "C"
