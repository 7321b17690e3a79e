// Base-field arithmetic of secp256k1 on nine 29-bit limbs, for the accumulation loops (fixed-base multi-exponentiation,
// MSM buckets): acc <- acc + q, q an affine table entry or input point.
//
// Why a second representation.  On gfx950 every VALU instruction costs about one issue slot, v_mad_u64_u32 included, so
// a modular multiplication is bounded by its instruction count.  With saturated 32-bit limbs (fe.h) each
// of the 64 partial products is followed by a carry into a third accumulator word: ~230 instructions.  With 29-bit
// limbs a whole column of nine products fits one 64-bit accumulator, so the product is 81 v_mad_u64_u32 with NO carry
// instructions; the eight high columns fold back through 2^261 = 2^37 + 31264 (mod p) as four more multiply-adds each,
// straight from their two registers; one carry pass (add / and / shift per limb) brings the nine low columns back to
// limbs: ~150 instructions.  Additions and subtractions are limb-wise with no carries: limbs may grow past 2^29
// between multiplications ("lazy" values), and a subtraction adds a multiple of p whose limbs dominate the subtrahend.
//
// Value classes (bound on every limb):
//   N   output of a reduction or of fe29_norm: < 2^29 + 2^18
//   L   one lazy step (N + (64 p - N), ...): < 2^30.6
// A column of nine products plus the fold's additions must stay below 2^64: N x N, N x L and (N x L + 2^30 x N) fit,
// L x L does not, so a difference is normalised before it is squared.
//
// The group law is total in the reference (`<>` of elliptic-curve-0.3.0; Utils.hs:71-85): the fast formula below is
// wrong for P + P, P + (-P) and an accumulator at infinity, all of which make ZZ3 = ZZ * PP = 0 (mod p).  A multiple of
// p below 2^262 has bits 38..255 all equal, so limbs 6 and 7 of such a ZZ3 are equal and all zeros or all ones whatever
// the small lazy excess of the low limbs would ripple to: that two-compare filter (no false negatives, false
// positives ~2^-57) sends those cases to the exact routine of ec.h.
#pragma once
#include <type_traits>
#include "ec.h"

namespace bp {

struct fe29 {
  uint32_t v[9];
};
struct xyzz29 {
  fe29 X, Y, ZZ, ZZZ;  // class N; the identity is ZZ = 0 (all limbs)
};

constexpr uint32_t M29 = 0x1FFFFFFFu;

// acc += a * b as ONE v_mad_u64_u32 on the device, accumulator in place (from C the compiler turns the power-of-two
// factors of the fold into 64-bit shift / add pairs and shuffles accumulator pairs)
BP_HD void mad29(uint64_t& acc, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE29_PORTABLE)
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
#else
  acc += (uint64_t)a * b;
#endif
}
// the same with a constant factor (kept in a scalar register)
BP_HD void mad29k(uint64_t& acc, uint32_t a, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE29_PORTABLE)
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
#else
  acc += (uint64_t)a * k;
#endif
}
BP_HD uint64_t mul29(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE29_PORTABLE)
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b) : "vcc");
  return d;
#else
  return (uint64_t)a * b;
#endif
}

// N products into one column as ONE asm block: a[0] b[0] + a[1] b[-1] + ... (the compiler puts a wait state behind every
// asm statement, so one statement per column costs 25 of them per multiplication instead of 113).  dot29z starts the
// column, dot29 adds to it.
template <int N>
BP_HD uint64_t dot29z(const uint32_t* a, const uint32_t* b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE29_PORTABLE)
  uint64_t d;
  if constexpr (N == 1)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        : "=&v"(d) : "v"(a[0]), "v"(b[0]) : "vcc");
  else if constexpr (N == 2)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        : "=&v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]) : "vcc");
  else if constexpr (N == 3)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        : "=&v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]) : "vcc");
  else if constexpr (N == 4)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        : "=&v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]) : "vcc");
  else if constexpr (N == 5)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        : "=&v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]) : "vcc");
  else if constexpr (N == 6)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        : "=&v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]), "v"(a[5]), "v"(b[-5]) : "vcc");
  else if constexpr (N == 7)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
        : "=&v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]), "v"(a[5]), "v"(b[-5]), "v"(a[6]), "v"(b[-6]) : "vcc");
  else if constexpr (N == 8)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
        : "=&v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]), "v"(a[5]), "v"(b[-5]), "v"(a[6]), "v"(b[-6]), "v"(a[7]), "v"(b[-7]) : "vcc");
  else if constexpr (N == 9)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
        : "=&v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]), "v"(a[5]), "v"(b[-5]), "v"(a[6]), "v"(b[-6]), "v"(a[7]), "v"(b[-7]), "v"(a[8]), "v"(b[-8]) : "vcc");
  return d;
#else
  uint64_t d = 0;
  for (int t = 0; t < N; t++) d += (uint64_t)a[t] * b[-t];
  return d;
#endif
}
template <int N>
BP_HD void dot29(uint64_t& d, const uint32_t* a, const uint32_t* b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE29_PORTABLE)
  if constexpr (N == 1)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        : "+v"(d) : "v"(a[0]), "v"(b[0]) : "vcc");
  else if constexpr (N == 2)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        : "+v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]) : "vcc");
  else if constexpr (N == 3)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        : "+v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]) : "vcc");
  else if constexpr (N == 4)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        : "+v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]) : "vcc");
  else if constexpr (N == 5)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        : "+v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]) : "vcc");
  else if constexpr (N == 6)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        : "+v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]), "v"(a[5]), "v"(b[-5]) : "vcc");
  else if constexpr (N == 7)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
        : "+v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]), "v"(a[5]), "v"(b[-5]), "v"(a[6]), "v"(b[-6]) : "vcc");
  else if constexpr (N == 8)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
        : "+v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]), "v"(a[5]), "v"(b[-5]), "v"(a[6]), "v"(b[-6]), "v"(a[7]), "v"(b[-7]) : "vcc");
  else if constexpr (N == 9)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
        : "+v"(d) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[-1]), "v"(a[2]), "v"(b[-2]), "v"(a[3]), "v"(b[-3]), "v"(a[4]), "v"(b[-4]), "v"(a[5]), "v"(b[-5]), "v"(a[6]), "v"(b[-6]), "v"(a[7]), "v"(b[-7]), "v"(a[8]), "v"(b[-8]) : "vcc");
#else
  for (int t = 0; t < N; t++) d += (uint64_t)a[t] * b[-t];
#endif
}

// 64 p as limbs in [2^30 - 2^16, 2^30): a - b = a + (BIAS - b) for b of class N; twice that for b < 2^30.9
BP_HD uint32_t fe29_bias(int i) {
  constexpr uint32_t B[9] = {0x3FFF0BC0u, 0x3FFFFDFEu, 0x3FFFFFFEu, 0x3FFFFFFEu, 0x3FFFFFFEu,
                             0x3FFFFFFEu, 0x3FFFFFFEu, 0x3FFFFFFEu, 0x3FFFFFFEu};
  return B[i];
}

// 8 x 32 (any value below 2^256) -> 9 x 29, exact limbs (limb 8 holds 24 bits)
BP_HD fe29 fe29_from_u256(const u256& a) {
  fe29 r;
  r.v[0] = a.v[0] & M29;
  r.v[1] = ((a.v[0] >> 29) | (a.v[1] << 3)) & M29;
  r.v[2] = ((a.v[1] >> 26) | (a.v[2] << 6)) & M29;
  r.v[3] = ((a.v[2] >> 23) | (a.v[3] << 9)) & M29;
  r.v[4] = ((a.v[3] >> 20) | (a.v[4] << 12)) & M29;
  r.v[5] = ((a.v[4] >> 17) | (a.v[5] << 15)) & M29;
  r.v[6] = ((a.v[5] >> 14) | (a.v[6] << 18)) & M29;
  r.v[7] = ((a.v[6] >> 11) | (a.v[7] << 21)) & M29;
  r.v[8] = a.v[7] >> 8;
  return r;
}

// any class -> 8 x 32, weakly reduced (< 2^256, possibly >= p), the form fe.h works on
BP_HD u256 fe29_to_u256(const fe29& a) {
  uint64_t l[9];
#pragma unroll
  for (int i = 0; i < 9; i++) l[i] = a.v[i];
  // carry pass, fold everything from bit 256 up (2^256 = 2^32 + 977; bit 256 is bit 24 of limb 8), three times
#pragma unroll
  for (int round = 0; round < 3; round++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      l[i + 1] += l[i] >> 29;
      l[i] &= M29;
    }
    uint64_t top = l[8] >> 24;
    l[8] &= 0xFFFFFFu;
    l[0] += top * 977u;
    l[1] += top << 3;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    l[i + 1] += l[i] >> 29;
    l[i] &= M29;
  }
  u256 r;
  r.v[0] = (uint32_t)(l[0] | (l[1] << 29));
  r.v[1] = (uint32_t)((l[1] >> 3) | (l[2] << 26));
  r.v[2] = (uint32_t)((l[2] >> 6) | (l[3] << 23));
  r.v[3] = (uint32_t)((l[3] >> 9) | (l[4] << 20));
  r.v[4] = (uint32_t)((l[4] >> 12) | (l[5] << 17));
  r.v[5] = (uint32_t)((l[5] >> 15) | (l[6] << 14));
  r.v[6] = (uint32_t)((l[6] >> 18) | (l[7] << 11));
  r.v[7] = (uint32_t)((l[7] >> 21) | (l[8] << 8));
  return r;
}

BP_HD fe29 fe29_zero() {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = 0;
  return r;
}
BP_HD fe29 fe29_add(const fe29& a, const fe29& b) {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
// 64 p - b  (b of class N): the lazy negative, < 2^30
BP_HD fe29 fe29_neg(const fe29& b) {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = fe29_bias(i) - b.v[i];
  return r;
}
// 128 p - b  (b < 2^30.9), < 2^31
BP_HD fe29 fe29_neg2(const fe29& b) {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = 2u * fe29_bias(i) - b.v[i];
  return r;
}

// limbs < 2^32 -> class N (same value mod p): one carry pass; what leaves limb 8 stands at 2^261 = 2^37 + 31264
BP_HD fe29 fe29_norm(const fe29& a) {
  fe29 r;
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint32_t t = a.v[i] + carry;  // < 2^32: callers pass limbs < 2^31
    r.v[i] = t & M29;
    carry = t >> 29;
  }
  r.v[0] += carry * 31264u;  // carry <= 7: < 2^18
  r.v[1] += carry << 8;
  return r;
}

// Reduction of 17 product columns (each < 2^63.9) to class N.
BP_HD fe29 fe29_reduce(uint64_t (&c)[17]) {
  // 2^(29 k) = 2^(29 (k - 9)) (2^37 + 31264) for k >= 9.  A high column c = lo + 2^32 hi first hands hi to the column
  // above it (2^32 2^(29 k) = 8 * 2^(29 (k + 1)); what arrives is below 2^35), so that columns 9..15 fold with two
  // multiply-adds each: lo * 31264 -> column k-9, lo * 2^8 -> column k-8.  Column 16 keeps what it was handed and
  // folds its hi as well: hi * 2^3 * 31264 -> column 8, and hi * 2^11 of column 9 once more, hi * 2^11 * 31264 ->
  // column 0, hi * 2^19 -> column 1.  26 multiply-adds (3 x 7 + 5) against the 32 of folding lo and hi of every column.
#pragma unroll
  for (int k = 9; k < 16; k++) {
    const uint32_t lo = (uint32_t)c[k], hi = (uint32_t)(c[k] >> 32);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE29_PORTABLE)
    asm("v_mad_u64_u32 %2, vcc, %4, 8, %2\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %5, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %3, %6, %1\n\t"
        : "+v"(c[k - 9]), "+v"(c[k - 8]), "+v"(c[k + 1])
        : "v"(lo), "v"(hi), "s"(31264u), "s"(256u)
        : "vcc");
#else
    mad29k(c[k + 1], hi, 8u);
    mad29k(c[k - 9], lo, 31264u);
    mad29k(c[k - 8], lo, 256u);
#endif
  }
  {
    const uint32_t lo = (uint32_t)c[16], hi = (uint32_t)(c[16] >> 32);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE29_PORTABLE)
    asm("v_mad_u64_u32 %0, vcc, %5, %9, %0\n\t"
        "v_mad_u64_u32 %2, vcc, %4, %6, %2\n\t"
        "v_mad_u64_u32 %1, vcc, %5, %10, %1\n\t"
        "v_mad_u64_u32 %3, vcc, %4, %7, %3\n\t"
        "v_mad_u64_u32 %3, vcc, %5, %8, %3\n\t"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[7]), "+v"(c[8])
        : "v"(lo), "v"(hi), "s"(31264u), "s"(256u), "s"(250112u), "s"(64028672u), "s"(524288u)
        : "vcc");
#else
    mad29k(c[7], lo, 31264u);
    mad29k(c[8], lo, 256u);
    mad29k(c[8], hi, 250112u);
    mad29k(c[0], hi, 64028672u);
    mad29k(c[1], hi, 524288u);
#endif
  }
  fe29 r;
  uint64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint64_t t = c[i] + carry;
    r.v[i] = (uint32_t)t & M29;
    carry = t >> 29;
  }
  // carry < 2^36 stands at 2^261 = 2^37 + 31264: clo * 31264 < 2^47 -> limbs 0, 1; clo * 2^8 -> limbs 1, 2; the four bits
  // chi stand at 2^293 = 2^32 (2^37 + 31264): chi * 2^11 at limb 2, chi * 2^3 * 31264 at limb 1
  const uint32_t clo = (uint32_t)carry, chi = (uint32_t)(carry >> 32);
  const uint64_t u = (uint64_t)clo * 31264u;
  uint32_t l0 = r.v[0] + ((uint32_t)u & M29);                                            // < 2^30
  uint32_t l1 = r.v[1] + (uint32_t)(u >> 29) + ((clo << 8) & M29) + chi * 250112u;       // < 2^30.1
  uint32_t l2 = r.v[2] + (clo >> 21) + (chi << 11);
  // two more carries keep class N tight (limb 2 ends below 2^29 + 2^16)
  l1 += l0 >> 29;
  l2 += l1 >> 29;
  r.v[0] = l0 & M29;
  r.v[1] = l1 & M29;
  r.v[2] = l2;
  return r;
}

// the 17 columns of a * b: started (FIRST) or added to.  Columns k and k + 9 hold nine products together; their
// multiply-adds alternate inside ONE asm block, so that a dependent v_mad_u64_u32 never directly follows its producer
// and the compiler's wait state behind every asm statement is paid 9 times per product instead of 81.
template <bool FIRST>
BP_HD void fe29_columns(uint64_t (&c)[17], const fe29& a, const fe29& b, std::integral_constant<bool, FIRST>) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE29_PORTABLE)
  if (FIRST) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %6, %7, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %10, %11, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %14, %15, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %18, %19, %1\n\t"
        : "=&v"(c[0]), "=&v"(c[9])
        : "v"(a.v[0]), "v"(b.v[0]), "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %10, %11, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %14, %15, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %18, %19, %1\n\t"
        : "=&v"(c[1]), "=&v"(c[10])
        : "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[0]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %14, %15, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %18, %19, %1\n\t"
        : "=&v"(c[2]), "=&v"(c[11])
        : "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[0]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %18, %19, %1\n\t"
        : "=&v"(c[3]), "=&v"(c[12])
        : "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[0]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
        : "=&v"(c[4]), "=&v"(c[13])
        : "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[0])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
        : "=&v"(c[5]), "=&v"(c[14])
        : "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
        : "=&v"(c[6]), "=&v"(c[15])
        : "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
        : "=&v"(c[7]), "=&v"(c[16])
        : "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
        : "=&v"(c[8])
        : "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0])
        : "vcc");
  }
  if (!FIRST) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %6, %7, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %10, %11, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %14, %15, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %18, %19, %1\n\t"
        : "+v"(c[0]), "+v"(c[9])
        : "v"(a.v[0]), "v"(b.v[0]), "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %10, %11, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %14, %15, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %18, %19, %1\n\t"
        : "+v"(c[1]), "+v"(c[10])
        : "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[0]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %14, %15, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %18, %19, %1\n\t"
        : "+v"(c[2]), "+v"(c[11])
        : "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[0]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %1, vcc, %18, %19, %1\n\t"
        : "+v"(c[3]), "+v"(c[12])
        : "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[0]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %16, %17, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
        : "+v"(c[4]), "+v"(c[13])
        : "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[0])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
        : "+v"(c[5]), "+v"(c[14])
        : "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
        : "+v"(c[6]), "+v"(c[15])
        : "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %18, %19, %0\n\t"
        : "+v"(c[7]), "+v"(c[16])
        : "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0])
        : "vcc");
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t"
        "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
        : "+v"(c[8])
        : "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0])
        : "vcc");
  }
#else
#define BP_COL(K, N, I0)                                        \
  if (FIRST) c[K] = dot29z<N>(&a.v[I0], &b.v[(K) - (I0)]);      \
  else dot29<N>(c[K], &a.v[I0], &b.v[(K) - (I0)]);
  BP_COL(0, 1, 0) BP_COL(1, 2, 0) BP_COL(2, 3, 0) BP_COL(3, 4, 0) BP_COL(4, 5, 0) BP_COL(5, 6, 0) BP_COL(6, 7, 0)
  BP_COL(7, 8, 0) BP_COL(8, 9, 0) BP_COL(9, 8, 1) BP_COL(10, 7, 2) BP_COL(11, 6, 3) BP_COL(12, 5, 4) BP_COL(13, 4, 5)
  BP_COL(14, 3, 6) BP_COL(15, 2, 7) BP_COL(16, 1, 8)
#undef BP_COL
#endif
}

// a * b (one operand may be lazy)
BP_HD fe29 fe29_mul(const fe29& a, const fe29& b) {
  uint64_t c[17];
  fe29_columns(c, a, b, std::integral_constant<bool, true>());
  return fe29_reduce(c);
}

// a * b + c * d in one reduction
BP_HD fe29 fe29_mul2(const fe29& a, const fe29& b, const fe29& cc, const fe29& d) {
  uint64_t c[17];
  fe29_columns(c, a, b, std::integral_constant<bool, true>());
  fe29_columns(c, cc, d, std::integral_constant<bool, false>());
  return fe29_reduce(c);
}

// a * b + e in one reduction: e (limbs < 2^32, e.g. the lazy negative of a class-N value) joins the low columns, so a
// difference U - X of a fresh product and a stored value comes out of the reduction in class N - no separate
// addition and normalisation (9 multiply-adds by one instead of ~40 instructions)
BP_HD fe29 fe29_mul_add(const fe29& a, const fe29& b, const fe29& e) {
  uint64_t c[17];
  fe29_columns(c, a, b, std::integral_constant<bool, true>());
#pragma unroll
  for (int i = 0; i < 9; i++) mad29k(c[i], e.v[i], 1u);
  return fe29_reduce(c);
}

// a * a + e for a of class N and a lazy e (limbs < 2^32) added to the low columns
BP_HD fe29 fe29_sqr_add(const fe29& a, const fe29& e) {
  uint32_t d[9];
#pragma unroll
  for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
  uint64_t c[17];
#pragma unroll
  for (int k = 0; k < 17; k++) c[k] = k < 9 ? (uint64_t)e.v[k] : 0;
  // cross terms 2 a_i a_j, i < j, i + j = k: one block per column
#define BP_SQC(K, N, I0) dot29<N>(c[K], &d[I0], &a.v[(K) - (I0)]);
  BP_SQC(1, 1, 0) BP_SQC(2, 1, 0) BP_SQC(3, 2, 0) BP_SQC(4, 2, 0) BP_SQC(5, 3, 0) BP_SQC(6, 3, 0) BP_SQC(7, 4, 0)
  BP_SQC(8, 4, 0) BP_SQC(9, 4, 1) BP_SQC(10, 3, 2) BP_SQC(11, 3, 3) BP_SQC(12, 2, 4) BP_SQC(13, 2, 5) BP_SQC(14, 1, 6)
  BP_SQC(15, 1, 7)
#undef BP_SQC
#pragma unroll
  for (int i = 0; i < 9; i++) mad29(c[2 * i], a.v[i], a.v[i]);
  return fe29_reduce(c);
}
BP_HD fe29 fe29_sqr(const fe29& a) { return fe29_sqr_add(a, fe29_zero()); }
// 3 a^2 for a of class N: the columns of the square (below 2^61.2) tripled before the reduction
BP_HD fe29 fe29_sqr_x3(const fe29& a) {
  uint32_t d[9];
#pragma unroll
  for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
  uint64_t c[17];
#pragma unroll
  for (int k = 0; k < 17; k++) c[k] = 0;
#define BP_SQC(K, N, I0) dot29<N>(c[K], &d[I0], &a.v[(K) - (I0)]);
  BP_SQC(1, 1, 0) BP_SQC(2, 1, 0) BP_SQC(3, 2, 0) BP_SQC(4, 2, 0) BP_SQC(5, 3, 0) BP_SQC(6, 3, 0) BP_SQC(7, 4, 0)
  BP_SQC(8, 4, 0) BP_SQC(9, 4, 1) BP_SQC(10, 3, 2) BP_SQC(11, 3, 3) BP_SQC(12, 2, 4) BP_SQC(13, 2, 5) BP_SQC(14, 1, 6)
  BP_SQC(15, 1, 7)
#undef BP_SQC
#pragma unroll
  for (int i = 0; i < 9; i++) mad29(c[2 * i], a.v[i], a.v[i]);
#pragma unroll
  for (int k = 0; k < 17; k++) c[k] += c[k] << 1;
  return fe29_reduce(c);
}

// a^(2^n)
BP_HD fe29 fe29_sqr_n(fe29 a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; i++) a = fe29_sqr(a);
  return a;
}
// 1/a mod p as a^(p-2) by the standard secp256k1 addition chain (255 squarings, 15 products) on 29-bit limbs: ~37 k
// issue slots against ~66 k for the almost-inverse of fe.h.  0 -> 0.  In and out: 8 x 32 (weakly reduced).
BP_HD u256 fq_inv29(const u256& a8) {
  const fe29 a = fe29_from_u256(a8);
  const fe29 x2 = fe29_mul(fe29_sqr(a), a);
  const fe29 x3 = fe29_mul(fe29_sqr(x2), a);
  const fe29 x6 = fe29_mul(fe29_sqr_n(x3, 3), x3);
  const fe29 x9 = fe29_mul(fe29_sqr_n(x6, 3), x3);
  const fe29 x11 = fe29_mul(fe29_sqr_n(x9, 2), x2);
  const fe29 x22 = fe29_mul(fe29_sqr_n(x11, 11), x11);
  const fe29 x44 = fe29_mul(fe29_sqr_n(x22, 22), x22);
  const fe29 x88 = fe29_mul(fe29_sqr_n(x44, 44), x44);
  const fe29 x176 = fe29_mul(fe29_sqr_n(x88, 88), x88);
  const fe29 x220 = fe29_mul(fe29_sqr_n(x176, 44), x44);
  const fe29 x223 = fe29_mul(fe29_sqr_n(x220, 3), x3);
  // p - 2 = 2^256 - 2^32 - 979: x223 | 0 | 22 ones | 0000 | 1 | 0 | 11 | 0 | 1
  fe29 t = fe29_mul(fe29_sqr_n(x223, 23), x22);
  t = fe29_mul(fe29_sqr_n(t, 5), a);
  t = fe29_mul(fe29_sqr_n(t, 3), x2);
  t = fe29_mul(fe29_sqr_n(t, 2), a);
  return fe29_to_u256(t);
}

BP_HD xyzz29 xyzz29_from(const xyzz& p) {
  xyzz29 r;
  r.X = fe29_from_u256(p.X);
  r.Y = fe29_from_u256(p.Y);
  r.ZZ = fe29_from_u256(fq_is_zero(p.ZZ) ? u256_zero() : p.ZZ);
  r.ZZZ = fe29_from_u256(p.ZZZ);
  return r;
}
BP_HD xyzz xyzz29_to(const xyzz29& p) {
  xyzz r;
  r.X = fe29_to_u256(p.X);
  r.Y = fe29_to_u256(p.Y);
  r.ZZ = fe29_to_u256(p.ZZ);
  r.ZZZ = fe29_to_u256(p.ZZZ);
  return r;
}
BP_HD xyzz29 xyzz29_inf() {
  xyzz29 r;
  r.X = fe29_zero();
  r.Y = fe29_zero();
  r.ZZ = fe29_zero();
  r.ZZZ = fe29_zero();
  return r;
}

// the identity as every routine here leaves it: ZZ = 0 in ALL limbs (the fast formulas send ZZ3 = 0 (mod p) to the exact
// routines, which return zero limbs)
BP_HD bool xyzz29_is_inf(const xyzz29& p) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= p.ZZ.v[i];
  return o == 0;
}

// 2 p (dbl-2008-s-1 with a = 0) on lazy limbs:
//   U = 2Y, V = U^2, W = U V, S = X V, M = 3 X^2, X3 = M^2 - 2S, Y3 = M (S - X3) - W Y, ZZ3 = V ZZ, ZZZ3 = W ZZZ
// as 4 products, 2 squares (one tripled in its columns), one square-and-add and one two-product reduction.  The identity
// (ZZ = 0) stays the identity: ZZ3 = V * 0; secp256k1 has no point with Y = 0.
BP_HD xyzz29 xyzz29_dbl(const xyzz29& p) {
  fe29 U;                                                     // 2Y < 2^30.001
#pragma unroll
  for (int i = 0; i < 9; i++) U.v[i] = p.Y.v[i] << 1;
  const fe29 V = fe29_sqr(U);                                 // cross terms 4 x 2^31 x 2^30 + 2^60: fits
  const fe29 W = fe29_mul(U, V);
  const fe29 S = fe29_mul(p.X, V);
  const fe29 M = fe29_sqr_x3(p.X);                            // 3 X^2, class N
  fe29 S2;
#pragma unroll
  for (int i = 0; i < 9; i++) S2.v[i] = S.v[i] << 1;
  xyzz29 r;
  r.X = fe29_sqr_add(M, fe29_neg2(S2));                       // M^2 + (128 p - 2S)
  const fe29 T = fe29_add(S, fe29_neg(r.X));                  // L
  r.Y = fe29_mul2(M, T, W, fe29_neg(p.Y));                    // M (S - X3) + W (64 p - Y)
  r.ZZ = fe29_mul(V, p.ZZ);
  r.ZZZ = fe29_mul(W, p.ZZZ);
  return r;
}

// p + q for two XYZZ points (add-2008-s) on lazy limbs; P + P, P - P and identities make ZZ3 = 0 and go to ec.h
BP_HD xyzz29 xyzz29_add_exact(const xyzz29& p, const xyzz29& q) {
  return xyzz29_from(xyzz_add(xyzz29_to(p), xyzz29_to(q)));
}
BP_HD xyzz29 xyzz29_add(const xyzz29& p, const xyzz29& q) {
  const fe29 U1 = fe29_mul(p.X, q.ZZ);
  const fe29 S1 = fe29_mul(p.Y, q.ZZZ);
  const fe29 nS1 = fe29_neg(S1);
  const fe29 Pd = fe29_mul_add(q.X, p.ZZ, fe29_neg(U1));      // U2 - U1, class N
  const fe29 Rd = fe29_mul_add(q.Y, p.ZZZ, nS1);              // S2 - S1
  const fe29 PP = fe29_sqr(Pd);
  const fe29 PPP = fe29_mul(Pd, PP);
  const fe29 Q = fe29_mul(U1, PP);
  fe29 W;
#pragma unroll
  for (int i = 0; i < 9; i++) W.v[i] = PPP.v[i] + (Q.v[i] << 1);
  xyzz29 r;
  r.X = fe29_sqr_add(Rd, fe29_neg2(W));
  const fe29 T = fe29_add(Q, fe29_neg(r.X));
  r.Y = fe29_mul2(Rd, T, nS1, PPP);
  r.ZZ = fe29_mul(fe29_mul(p.ZZ, q.ZZ), PP);
  r.ZZZ = fe29_mul(fe29_mul(p.ZZZ, q.ZZZ), PPP);
  const uint32_t l6 = r.ZZ.v[6], l7 = r.ZZ.v[7];
  if (l6 == l7 && (l6 == 0u || l6 == M29)) return xyzz29_add_exact(p, q);
  return r;
}

// (x, +-y, 1, 1), or the identity for q = O
BP_HD xyzz29 xyzz29_from_affine(const affine& q, bool negate) {
  if (affine_is_inf(q)) return xyzz29_inf();
  xyzz29 r;
  r.X = fe29_from_u256(q.x);
  r.Y = fe29_from_u256(negate ? fq_neg(q.y) : q.y);
  r.ZZ = fe29_zero();
  r.ZZ.v[0] = 1;
  r.ZZZ = r.ZZ;
  return r;
}

// The cases the lazy formula cannot do (P + P, P - P, acc = O), through ec.h.  Rare, and INLINE on purpose: as an
// out-of-line call it returned through memory, and the compiler then stored every fast-path result to the same
// stack slot as well (36 dwords of scratch traffic per mixed addition, 2.3 GB per k_fixed_multiexp launch).
BP_HD xyzz29 xyzz29_madd_exact(const xyzz29& p, const affine& q, bool negate) {
  affine qq = q;
  if (negate) qq = affine_neg(q);
  return xyzz29_from(xyzz_madd(xyzz29_to(p), qq));
}

// acc + (+-q), q a FINITE affine point below 2^256 per coordinate (a table entry, an input point): madd-2008-s on lazy limbs
//   U2 = x ZZ, S2 = y ZZZ, P = U2 - X, R = S2 - Y, PP = P^2, PPP = P PP, Q = X PP,
//   X3 = R^2 - PPP - 2Q, Y3 = R (Q - X3) - Y PPP, ZZ3 = ZZ PP, ZZZ3 = ZZZ PPP
// as 6 products (two of them with the subtrahend folded into their low columns), one square, one square-and-add and one
// two-product reduction.
BP_HD xyzz29 xyzz29_madd(const xyzz29& p, const affine& q, bool negate) {
  const fe29 qx = fe29_from_u256(q.x);
  fe29 qy = fe29_from_u256(q.y);
  if (negate) qy = fe29_neg(qy);                              // < 2^30
  const fe29 nY = fe29_neg(p.Y);                              // < 2^30
  const fe29 Pd = fe29_mul_add(qx, p.ZZ, fe29_neg(p.X));      // U2 - X out of one reduction: N
  const fe29 Rd = fe29_mul_add(qy, p.ZZZ, nY);                // S2 - Y (2^30 x N + 2^30): N
  const fe29 PP = fe29_sqr(Pd);
  const fe29 PPP = fe29_mul(Pd, PP);
  const fe29 Q = fe29_mul(p.X, PP);
  fe29 W;                                                     // PPP + 2Q < 2^30.6
#pragma unroll
  for (int i = 0; i < 9; i++) W.v[i] = PPP.v[i] + (Q.v[i] << 1);
  xyzz29 r;
  r.X = fe29_sqr_add(Rd, fe29_neg2(W));                       // R^2 + (128 p - W) -> N
  const fe29 T = fe29_add(Q, fe29_neg(r.X));                  // L
  r.Y = fe29_mul2(Rd, T, nY, PPP);                            // N x L + 2^30 x N -> N
  r.ZZ = fe29_mul(p.ZZ, PP);
  r.ZZZ = fe29_mul(p.ZZZ, PPP);
  const uint32_t l6 = r.ZZ.v[6], l7 = r.ZZ.v[7];
  // ZZ3 may be 0 (mod p): P + P, P - P or acc = O; the exact routine decides
  if (l6 == l7 && (l6 == 0u || l6 == M29)) return xyzz29_madd_exact(p, q, negate);
  return r;
}

}  // namespace bp
