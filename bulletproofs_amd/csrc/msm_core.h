// Pippenger bucket-MSM building blocks shared by the HIP kernels (msm.hip) and
// the host simulation used by the CPU tests (tests/hostsim).
//
// Replaces `sumExps` (/root/reference/Bulletproofs/Utils.hs:76-80): sum_i e_i*P_i.
// The reference evaluates N independent double-and-add products; the group
// element is the same whichever way the sum is associated, so the affine
// result is identical.
#pragma once
#include "ec.h"

namespace bp {

// Signed-digit window plan for 256-bit scalars.
//   digit_w in [-2^(c-1), 2^(c-1)], sum_w digit_w 2^(c w) = scalar
//   W = ceil(257 / c) windows: the top window has a spare bit, so no carry leaves it.
//   bucket index b = |digit| - 1 in [0, 2^(c-1)); digit 0 contributes nothing.
struct MsmPlan {
  int c;         // window bits, 1 <= c <= 16
  int W;         // windows
  int nb;        // buckets per window = 2^(c-1)
  int seg;       // buckets per reduction segment (power of two, <= nb)
  int nseg;      // segments per window = nb / seg
};

// `bits`: width of the scalars fed to the recoder (256 for plain scalars, 129 for GLV half-scalars);
// one more bit is kept for the carry of the signed recoding.
#ifndef MSM_SEG
#define MSM_SEG 8
#endif
BP_HD MsmPlan msm_plan(int c, int bits = 256) {
  MsmPlan p;
  p.c = c;
  p.W = (bits + 1 + c - 1) / c;
  p.nb = 1 << (c - 1);
  p.seg = p.nb < MSM_SEG ? p.nb : MSM_SEG;
  p.nseg = p.nb / p.seg;
  return p;
}

// Window size for N terms, measured on MI355X (benchmarks/msm_bench.py --c; ms per sum, GLV halves, two-level sort):
//   n          2^5   2^8   2^11  2^13  2^14  2^15  2^16  2^17
//   c = 13     0.70  0.71  0.78  0.90  1.02  1.18  1.47
//   c = 14                 0.87  0.93  0.95  0.96  0.99
//   c = 16                             0.94  0.99  0.99  0.98      (2^18: 1.15, 2^20: 2.09)
// c = 15 has the same nine GLV windows as c = 16 and half the buckets, but its top window holds 10 bits instead of 2:
// 512 buckets of ~4,000 entries each stretch k_accumulate from 1.00 to 1.59 ms at 2^20 while k_seg_reduce / k_final,
// latency chains whose depth does not depend on the bucket count, stay at 0.22 / 0.28 ms (2.54 against 1.98 ms per
// sum; 1.65 against 0.95 ms at 2^16.  profiles/r07_msm_window_15_ab.txt).
// Windows below 13 bits leave the two-level sort (FINE_BITS) and cost 0.9-2.9 ms at these sizes; fewer, wider windows
// also shorten the Horner chain.  Narrow windows remain reachable through `window_bits`.
inline int msm_choose_c(size_t n) {
  if (n < 16384) return 13;
  if (n < 131072) return 14;
  return 16;
}

constexpr uint32_t MSM_SKIP = 0xFFFFFFFFu;  // digit 0: no bucket
constexpr uint32_t MSM_NEG = 0x80000000u;   // entry flag: subtract the point

// Next signed digit of the scalar (least significant window first): consumes
// the low c bits of `s` (s >>= c) and the incoming carry.  Returns
// code = bucket | sign flag, or MSM_SKIP for a zero digit.  Static limb
// indices only, so the scalar stays in registers.
BP_HD uint32_t recode_next(u256& s, int c, uint32_t& carry) {
  uint32_t v = (s.v[0] & ((1u << c) - 1u)) + carry;
#pragma unroll
  for (int i = 0; i < 7; i++) s.v[i] = (s.v[i] >> c) | (s.v[i + 1] << (32 - c));
  s.v[7] >>= c;
  uint32_t half = 1u << (c - 1);
  if (v > half) {
    carry = 1;
    uint32_t mag = (1u << c) - v;  // 1 .. half-1 (or 0 when v == 2^c)
    return mag == 0 ? MSM_SKIP : ((mag - 1) | MSM_NEG);
  }
  carry = 0;
  return v == 0 ? MSM_SKIP : (v - 1);
}

// acc += (+/-) P
BP_HD void bucket_add(xyzz& acc, const affine& p, bool negate) {
  affine q = p;
  if (negate) q.y = fq_neg(p.y);  // (0,0) stays the identity: fq_neg(0) = 0
  acc = xyzz_madd(acc, q);
}

// Running-sum reduction of one segment of buckets B[0..seg):
//   S = sum_j B[j],  T = sum_j (j + 1) * B[j]
template <class LoadBucket>
BP_HD void segment_reduce(int seg, LoadBucket load, xyzz& S, xyzz& T) {
  S = xyzz_inf();
  T = xyzz_inf();
  for (int j = seg - 1; j >= 0; j--) {
    S = xyzz_add(S, load(j));
    T = xyzz_add(T, S);
  }
}

// k*P for a small k (< 2^bits), P in XYZZ
BP_HD xyzz xyzz_mul_small(const xyzz& p, uint32_t k, int bits) {
  xyzz acc = xyzz_inf();
  for (int i = bits - 1; i >= 0; i--) {
    acc = xyzz_dbl(acc);
    if ((k >> i) & 1u) acc = xyzz_add(acc, p);
  }
  return acc;
}

// Contribution of segment `sidx` (first bucket index sidx*seg, whose weight is
// sidx*seg + 1):  T + (sidx*seg) * S
BP_HD xyzz segment_value(const MsmPlan& pl, int sidx, const xyzz& S, const xyzz& T) {
  if (sidx == 0) return T;
  int seg_log = 0;
  while ((1 << seg_log) < pl.seg) seg_log++;
  int idx_bits = 1;  // sidx < nseg
  while ((1 << idx_bits) < pl.nseg) idx_bits++;
  xyzz m = xyzz_mul_small(S, (uint32_t)sidx, idx_bits);
  for (int i = 0; i < seg_log; i++) m = xyzz_dbl(m);
  return xyzz_add(T, m);
}

// Horner combine of the per-window sums: sum_w 2^(c w) * Wsum[w]
template <class LoadW>
BP_HD xyzz window_combine(const MsmPlan& pl, LoadW load) {
  xyzz acc = load(pl.W - 1);
  for (int w = pl.W - 2; w >= 0; w--) {
    for (int i = 0; i < pl.c; i++) acc = xyzz_dbl(acc);
    acc = xyzz_add(acc, load(w));
  }
  return acc;
}

}  // namespace bp
