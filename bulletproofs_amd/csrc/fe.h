// 256-bit modular arithmetic for secp256k1 on 8 x 32-bit limbs (gfx950 VALU).
//
// Replaces the arithmetic the reference imports from galois-field-1.0.0
// (`Fq`, `Fr`, `recip`, `^`; /root/reference/Bulletproofs/Utils.hs:8-9).
//
// Every 32x32 product below is written as (u64)a*b + c so that hipcc emits one
// v_mad_u64_u32 per limb product; everything is fully unrolled and lives in
// VGPRs.  The same source compiles for the host (tests/hostsim) so that the
// arithmetic can be checked against the oracle without a GPU.
//
//  * Fq  (base field, p = 2^256 - 2^32 - 977).  Values are kept WEAKLY reduced:
//    any 256-bit integer congruent to the element (so p+3 is a legal encoding
//    of 3).  Multiplication folds the high half with 2^256 = C (mod p),
//    C = 2^32 + 977.  fq_canon() gives the unique representative in [0, p).
//  * Mont<M> (generic 8-limb Montgomery CIOS).  Used for the scalar field Fr
//    (M = FrMod) and instantiable for Fq (M = FqMod) as the A/B arm of the
//    field microbenchmark.  Values are canonical (< modulus), in Montgomery
//    form (x * 2^256 mod m).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BP_HD __host__ __device__ __forceinline__
#else
#define BP_HD inline
#endif

namespace bp {

struct u256 {
  uint32_t v[8];
};

BP_HD u256 u256_zero() {
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = 0;
  return r;
}
BP_HD u256 u256_one() {
  u256 r = u256_zero();
  r.v[0] = 1;
  return r;
}
BP_HD bool u256_eq(const u256& a, const u256& b) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d |= a.v[i] ^ b.v[i];
  return d == 0;
}
BP_HD bool u256_is_zero(const u256& a) {
  uint32_t d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d |= a.v[i];
  return d == 0;
}
// r = a + b, returns carry out
BP_HD uint32_t u256_add(u256& r, const u256& a, const u256& b) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a.v[i] + b.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  return (uint32_t)c;
}
// r = a - b, returns borrow out (0/1)
BP_HD uint32_t u256_sub(u256& r, const u256& a, const u256& b) {
  int64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (int64_t)a.v[i] - (int64_t)b.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;  // arithmetic shift: 0 or -1
  }
  return (uint32_t)(c & 1);
}
// a >= b ?
BP_HD bool u256_geq(const u256& a, const u256& b) {
  u256 t;
  return u256_sub(t, a, b) == 0;
}
BP_HD uint32_t u256_bit(const u256& a, int i) { return (a.v[i >> 5] >> (i & 31)) & 1u; }
// Most significant bit first, without indexing limbs by a runtime value (keeps the value in
// registers on the GPU): returns bit 255 and shifts the value left by one.
BP_HD uint32_t u256_pop_msb(u256& a) {
  uint32_t top = a.v[7] >> 31;
#pragma unroll
  for (int i = 7; i > 0; i--) a.v[i] = (a.v[i] << 1) | (a.v[i - 1] >> 31);
  a.v[0] <<= 1;
  return top;
}

// acc(64) + ext(32) += a*b : one v_mad_u64_u32 plus one v_addc on the device.
BP_HD void mac96(uint64_t& acc, uint32_t& ext, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(acc), "+v"(ext)
      : "v"(a), "v"(b)
      : "vcc");
#else
  uint64_t p = (uint64_t)a * b;
  acc += p;
  ext += (acc < p) ? 1u : 0u;
#endif
}

// First product of a column: acc(64) += a*b and ext(32) = carry, without reading ext.
BP_HD void mac96_first(uint64_t& acc, uint32_t& ext, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e64 %1, vcc, 0, 0, vcc"
      : "+v"(acc), "=v"(ext)
      : "v"(a), "v"(b)
      : "vcc");
#else
  uint64_t p = (uint64_t)a * b;
  acc += p;
  ext = (acc < p) ? 1u : 0u;
#endif
}

// 8x8 -> 16 limb schoolbook product, row-wise (operand scanning).  Kept as the
// A/B arm of benchmarks/microbench.hip.
BP_HD void mul_wide_rows(uint32_t t[16], const u256& a, const u256& b) {
  uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    c += (uint64_t)a.v[0] * b.v[j];
    t[j] = (uint32_t)c;
    c >>= 32;
  }
  t[8] = (uint32_t)c;
#pragma unroll
  for (int i = 1; i < 8; i++) {
    c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c += (uint64_t)a.v[i] * b.v[j] + t[i + j];
      t[i + j] = (uint32_t)c;
      c >>= 32;
    }
    t[i + 8] = (uint32_t)c;
  }
}

// ---- carry chains as single asm blocks (device): one VALU instruction per limb, carry in VCC ----
#if defined(__HIP_DEVICE_COMPILE__)
// r[0..7] += s[0..7]; returns the carry out (0/1)
__device__ __forceinline__ uint32_t dev_add8(uint32_t* r, const uint32_t* s) {
  uint32_t co = 0;
  asm("v_add_co_u32 %0, vcc, %0, %9\n\t"
      "v_addc_co_u32 %1, vcc, %1, %10, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, %11, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %12, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %4, %13, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %14, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %6, %15, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %7, %16, vcc\n\t"
      "v_addc_co_u32 %8, vcc, 0, %8, vcc"
      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(co)
      : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7])
      : "vcc");
  return co;
}
// r[0] += x0, r[1] += x1, then the carry ripples through r[2..7]; returns the carry out
__device__ __forceinline__ uint32_t dev_add2_ripple8(uint32_t* r, uint32_t x0, uint32_t x1) {
  uint32_t co = 0;
  asm("v_add_co_u32 %0, vcc, %0, %9\n\t"
      "v_addc_co_u32 %1, vcc, %1, %10, vcc\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
      "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
      "v_addc_co_u32 %7, vcc, 0, %7, vcc\n\t"
      "v_addc_co_u32 %8, vcc, 0, %8, vcc"
      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(co)
      : "v"(x0), "v"(x1)
      : "vcc");
  return co;
}
// r[0] += x0, r[1] += x1, carry ripples through r[2..7] (no carry out kept: callers bound the value)
__device__ __forceinline__ void dev_add2_ripple8_nc(uint32_t* r, uint32_t x0, uint32_t x1) {
  asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
      "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
      "v_addc_co_u32 %7, vcc, 0, %7, vcc"
      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
      : "v"(x0), "v"(x1)
      : "vcc");
}
#endif

// 8x8 -> 16 limb product, column-wise (product scanning) into 96-bit accumulators: the 64-bit
// addend of v_mad_u64_u32 is the accumulator itself, so no zero-extension moves are needed, and
// the carry-out feeds one v_addc.  On the device the low columns (0..7) and the high columns
// (8..14) run as two interleaved, independent chains - a dependent v_mad_u64_u32 otherwise
// waits on its predecessor - and the low chain's final carry is rippled into the high limbs.
BP_HD void mul_wide(uint32_t t[16], const u256& a, const u256& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE_PORTABLE_MUL)
  // generated by benchmarks/gen_fe_schedule.py mul: low chain = columns 0..7 (36 products), high chain =
  // columns 8..14 (28 products), one product of each in turn, one asm block per run between column ends
  uint64_t accL = 0, accH = 0;
  uint32_t extL, extH;
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc"
      : "+v"(accL), "=&v"(extL)
      : "v"(a.v[0]), "v"(b.v[0])
      : "vcc");
  t[0] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %4, %10, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc"
      : "+v"(accH), "=&v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[6]), "v"(b.v[0])
      : "vcc");
  t[1] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %10, %11, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %14, %15, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc"
      : "+v"(accH), "+v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[0])
      : "vcc");
  t[2] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accH), "+v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[1])
      : "vcc");
  t[8] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %6, %8, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %9, %10, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %9, %11, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accL), "+v"(extL), "+v"(accH), "=&v"(extH)
      : "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[7]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[6]), "v"(b.v[0])
      : "vcc");
  t[3] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %7, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %9, %10, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %11, %10, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %12, %13, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %14, %13, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accH), "+v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[5]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[6]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[7])
      : "vcc");
  t[9] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %4, %6, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accL), "+v"(extL), "+v"(accH), "=&v"(extH)
      : "v"(a.v[3]), "v"(b.v[1]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[0])
      : "vcc");
  t[4] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %7, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %9, %10, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %11, %10, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %12, %13, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %14, %13, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accH), "+v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[6]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[7])
      : "vcc");
  t[10] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %6, %8, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %9, %10, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %9, %11, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accL), "+v"(extL), "+v"(accH), "=&v"(extH)
      : "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[7]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[6]), "v"(b.v[0])
      : "vcc");
  t[5] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accH), "+v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[4])
      : "vcc");
  t[11] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %10, %11, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %14, %5, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc"
      : "+v"(accL), "+v"(extL), "+v"(accH), "=&v"(extH)
      : "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[7])
      : "vcc");
  t[12] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %10, %11, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc"
      : "+v"(accL), "+v"(extL), "+v"(accH), "=&v"(extH)
      : "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[6])
      : "vcc");
  t[13] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accL), "+v"(extL)
      : "v"(a.v[6]), "v"(b.v[0])
      : "vcc");
  t[6] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc"
      : "+v"(accH), "=&v"(extH)
      : "v"(a.v[7]), "v"(b.v[7])
      : "vcc");
  t[14] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %10, %11, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %12, %13, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %16, %17, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accL), "=&v"(extL)
      : "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0])
      : "vcc");
  t[7] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  t[15] = (uint32_t)accH;
  dev_add2_ripple8_nc(t + 8, (uint32_t)accL, (uint32_t)(accL >> 32));  // the product is < 2^512: no carry out
#else
  uint64_t acc = 0;
  uint32_t ext = 0;
#pragma unroll
  for (int k = 0; k < 15; k++) {
    const int lo = k < 8 ? 0 : k - 7;
    const int hi = k < 8 ? k : 7;
#pragma unroll
    for (int i = lo; i <= hi; i++) mac96(acc, ext, a.v[i], b.v[k - i]);
    t[k] = (uint32_t)acc;
    acc = (acc >> 32) | ((uint64_t)ext << 32);
    ext = 0;
  }
  t[15] = (uint32_t)acc;
#endif
}

// 8 -> 16 limb square: off-diagonal products once, doubled, plus the diagonal.
BP_HD void sqr_wide(uint32_t t[16], const u256& a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE_PORTABLE_SQR)
  // off-diagonal column sums S_k = sum_{i<j, i+j=k} a_i a_j, k = 1..13, as two interleaved chains
  // (columns 1..7 and 8..13), exactly as in mul_wide
  // generated by benchmarks/gen_fe_schedule.py sqr
  uint64_t accL = 0, accH = 0;
  uint32_t extL, extH;
  uint32_t o[16];
  o[0] = 0;
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc"
      : "+v"(accL), "=&v"(extL)
      : "v"(a.v[0]), "v"(a.v[1])
      : "vcc");
  o[1] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc"
      : "+v"(accH), "=&v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[1]), "v"(a.v[7]), "v"(a.v[0]), "v"(a.v[2])
      : "vcc");
  o[2] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accH), "+v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[2]), "v"(a.v[6]), "v"(a.v[0]), "v"(a.v[3]), "v"(a.v[5])
      : "vcc");
  o[8] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accL), "+v"(extL)
      : "v"(a.v[1]), "v"(a.v[2])
      : "vcc");
  o[3] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %10, %8, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc"
      : "+v"(accH), "=&v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[2]), "v"(a.v[7]), "v"(a.v[0]), "v"(a.v[4]), "v"(a.v[3]), "v"(a.v[6]), "v"(a.v[1])
      : "vcc");
  o[4] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accH), "+v"(extH)
      : "v"(a.v[4]), "v"(a.v[5])
      : "vcc");
  o[9] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %9, %10, %2\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc"
      : "+v"(accL), "=&v"(extL), "+v"(accH), "=&v"(extH)
      : "v"(a.v[0]), "v"(a.v[5]), "v"(a.v[3]), "v"(a.v[7]), "v"(a.v[1]), "v"(a.v[4]), "v"(a.v[6])
      : "vcc");
  o[10] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accL), "+v"(extL)
      : "v"(a.v[2]), "v"(a.v[3])
      : "vcc");
  o[5] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %6, %7, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %7, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accH), "=&v"(extH), "+v"(accL), "=&v"(extL)
      : "v"(a.v[4]), "v"(a.v[7]), "v"(a.v[0]), "v"(a.v[6]), "v"(a.v[5])
      : "vcc");
  o[11] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %5, %6, %2\n\t"
      "v_addc_co_u32_e64 %3, vcc, 0, 0, vcc"
      : "+v"(accL), "+v"(extL), "+v"(accH), "=&v"(extH)
      : "v"(a.v[1]), "v"(a.v[5]), "v"(a.v[7])
      : "vcc");
  o[12] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accL), "+v"(extL)
      : "v"(a.v[2]), "v"(a.v[4])
      : "vcc");
  o[6] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc"
      : "+v"(accH), "=&v"(extH)
      : "v"(a.v[6]), "v"(a.v[7])
      : "vcc");
  o[13] = (uint32_t)accH;
  accH = (accH >> 32) | ((uint64_t)extH << 32);
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, 0, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %6, %7, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(accL), "=&v"(extL)
      : "v"(a.v[0]), "v"(a.v[7]), "v"(a.v[1]), "v"(a.v[6]), "v"(a.v[2]), "v"(a.v[5]), "v"(a.v[3]), "v"(a.v[4])
      : "vcc");
  o[7] = (uint32_t)accL;
  accL = (accL >> 32) | ((uint64_t)extL << 32);
  o[14] = (uint32_t)accH;
  o[15] = (uint32_t)(accH >> 32);
  dev_add2_ripple8_nc(o + 8, (uint32_t)accL, (uint32_t)(accL >> 32));
  // double
#pragma unroll
  for (int i = 15; i > 0; i--) t[i] = (o[i] << 1) | (o[i - 1] >> 31);
  t[0] = 0;
  // add the diagonal squares (independent products), as two 8-limb carry chains
  uint32_t d[16];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t sq = (uint64_t)a.v[i] * a.v[i];
    d[2 * i] = (uint32_t)sq;
    d[2 * i + 1] = (uint32_t)(sq >> 32);
  }
  uint32_t c = dev_add8(t, d);
  d[8] += c;  // the low word of a square is never 2^32 - 1 (-1 is not a square mod 4): no overflow
  (void)dev_add8(t + 8, d + 8);  // a^2 < 2^512: no carry out
#else
  uint64_t c;
  t[0] = 0;
  // off-diagonal: sum_{i<j} a_i a_j 2^(32(i+j)), rows i = 0..6
  c = 0;
#pragma unroll
  for (int j = 1; j < 8; j++) {
    c += (uint64_t)a.v[0] * a.v[j];
    t[j] = (uint32_t)c;
    c >>= 32;
  }
  t[8] = (uint32_t)c;
#pragma unroll
  for (int i = 1; i < 7; i++) {
    c = 0;
#pragma unroll
    for (int j = i + 1; j < 8; j++) {
      c += (uint64_t)a.v[i] * a.v[j] + t[i + j];
      t[i + j] = (uint32_t)c;
      c >>= 32;
    }
    t[i + 8] = (uint32_t)c;
  }
  t[15] = 0;
  // double
#pragma unroll
  for (int i = 15; i > 0; i--) t[i] = (t[i] << 1) | (t[i - 1] >> 31);
  // add the diagonal squares
  c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a.v[i] * a.v[i] + t[2 * i];
    t[2 * i] = (uint32_t)c;
    c >>= 32;
    c += t[2 * i + 1];
    t[2 * i + 1] = (uint32_t)c;
    c >>= 32;
  }
#endif
}

// ----------------------------------------------------------------------------
// Fq: p = 2^256 - C, C = 2^32 + 977
// ----------------------------------------------------------------------------
BP_HD u256 fq_p() {
  u256 p;
  p.v[0] = 0xFFFFFC2Fu;
  p.v[1] = 0xFFFFFFFEu;
#pragma unroll
  for (int i = 2; i < 8; i++) p.v[i] = 0xFFFFFFFFu;
  return p;
}

// r += k * C for a small k (0..3), r known to be tiny after a wrap past 2^256: touches 3 limbs.
BP_HD void fq_fold_small(u256& r, uint32_t k) {
  uint64_t c = (uint64_t)r.v[0] + k * 977u;
  r.v[0] = (uint32_t)c;
  c >>= 32;
  c += (uint64_t)r.v[1] + k;
  r.v[1] = (uint32_t)c;
  c >>= 32;
  r.v[2] += (uint32_t)c;
}

BP_HD u256 fq_add(const u256& a, const u256& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE_PORTABLE_ADDSUB)
  // a + b, then + C where the sum wrapped past 2^256 (2^256 = C mod p), then the rare second wrap on three limbs: 23 instructions
  u256 r;
  uint32_t k0, k1;
  asm(
      "v_add_co_u32 %0, vcc, %10, %18\n\t"
      "v_addc_co_u32 %1, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %15, %23, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %16, %24, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %17, %25, vcc\n\t"
      "v_cndmask_b32 %8, 0, %26, vcc\n\t"
      "v_addc_co_u32_e64 %9, vcc, 0, 0, vcc\n\t"
      "v_add_co_u32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
      "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
      "v_addc_co_u32 %7, vcc, 0, %7, vcc\n\t"
      "v_cndmask_b32 %8, 0, %26, vcc\n\t"
      "v_addc_co_u32_e64 %9, vcc, 0, 0, vcc\n\t"
      "v_add_co_u32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc"
      : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&v"(k0), "=&v"(k1)
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]), "v"(977u)
      : "vcc");
  return r;
#else
  u256 r;
  uint32_t cy = u256_add(r, a, b);
  // 2^256 = C (mod p): add cy*C with full ripple, then a rare second wrap
  uint64_t c = (uint64_t)r.v[0] + (cy ? 977u : 0u);
  r.v[0] = (uint32_t)c;
  c >>= 32;
  c += (uint64_t)r.v[1] + cy;
  r.v[1] = (uint32_t)c;
  c >>= 32;
#pragma unroll
  for (int i = 2; i < 8; i++) {
    c += r.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  fq_fold_small(r, (uint32_t)c);
  return r;
#endif
}

BP_HD u256 fq_sub(const u256& a, const u256& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE_PORTABLE_ADDSUB)
  // a - b, then - C where it borrowed (the wrapped value is a - b + 2^256 = a - b + C mod p), then the rare second borrow on two limbs: 22 instructions
  u256 r;
  uint32_t k0, k1;
  asm(
      "v_sub_co_u32 %0, vcc, %10, %18\n\t"
      "v_subb_co_u32 %1, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %16, %24, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %17, %25, vcc\n\t"
      "v_cndmask_b32 %8, 0, %26, vcc\n\t"
      "v_addc_co_u32_e64 %9, vcc, 0, 0, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %0, %8\n\t"
      "v_subb_co_u32 %1, vcc, %1, %9, vcc\n\t"
      "v_subbrev_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "v_subbrev_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_subbrev_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "v_subbrev_co_u32 %5, vcc, 0, %5, vcc\n\t"
      "v_subbrev_co_u32 %6, vcc, 0, %6, vcc\n\t"
      "v_subbrev_co_u32 %7, vcc, 0, %7, vcc\n\t"
      "v_cndmask_b32 %8, 0, %26, vcc\n\t"
      "v_addc_co_u32_e64 %9, vcc, 0, 0, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %0, %8\n\t"
      "v_subb_co_u32 %1, vcc, %1, %9, vcc"
      : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]), "=&v"(k0), "=&v"(k1)
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]), "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7]), "v"(977u)
      : "vcc");
  return r;
#else
  u256 r;
  uint32_t bw = u256_sub(r, a, b);
  // wrapped value is a-b+2^256 = a-b+C (mod p): subtract bw*C
  int64_t c = (int64_t)r.v[0] - (bw ? 977 : 0);
  r.v[0] = (uint32_t)c;
  c >>= 32;
  c += (int64_t)r.v[1] - (int64_t)bw;
  r.v[1] = (uint32_t)c;
  c >>= 32;
#pragma unroll
  for (int i = 2; i < 8; i++) {
    c += (int64_t)r.v[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  // rare second borrow: value is then 2^256 - e with e <= C; only limbs 0,1 move
  uint32_t bw2 = (uint32_t)(c & 1);
  c = (int64_t)r.v[0] - (bw2 ? 977 : 0);
  r.v[0] = (uint32_t)c;
  c >>= 32;
  r.v[1] = (uint32_t)((int64_t)r.v[1] - (int64_t)bw2 + c);
  return r;
#endif
}

BP_HD u256 fq_neg(const u256& a) { return fq_sub(u256_zero(), a); }
BP_HD u256 fq_dbl(const u256& a) { return fq_add(a, a); }

// Fold a 512-bit product to a weakly reduced 256-bit value.
BP_HD u256 fq_reduce_wide(const uint32_t t[16]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BP_FE_PORTABLE_REDUCE)
  // lo + hi * C, C = 2^32 + 977: eight independent products hi_i * 977, then three carry chains
  uint32_t a[9], plo[8], phi[8], hi[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t pr = (uint64_t)t[8 + i] * 977u;
    plo[i] = (uint32_t)pr;
    phi[i] = (uint32_t)(pr >> 32);  // < 2^10
    a[i] = t[i];
    hi[i] = t[8 + i];
  }
  a[8] = dev_add8(a, plo);          // a[0..8] = lo + sum plo_i 2^(32 i)
  (void)dev_add8(a + 1, phi);       // + sum phi_i 2^(32 (i+1)): a[8] <= 2^10 + 1, no carry out
  uint32_t a9 = dev_add8(a + 1, hi);  // + hi << 32
  // second fold: h2 = a9:a[8] < 2^33; add h2 * 977 at limb 0 and h2 at limb 1
  uint64_t w = (uint64_t)a[8] * 977u + (a9 ? 977ull << 32 : 0ull);
  u256 r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = a[i];
  uint32_t k = dev_add2_ripple8(r.v, (uint32_t)w, (uint32_t)(w >> 32));
  uint32_t r1[7];
#pragma unroll
  for (int i = 0; i < 7; i++) r1[i] = r.v[i + 1];
  // r.v[1..7] += (a[8], a9) with ripple: reuse the 8-limb chain on limbs 1..7 plus a scratch top
  {
    uint32_t co = 0;
    asm("v_add_co_u32 %0, vcc, %0, %8\n\t"
        "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
        "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
        "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
        "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
        "v_addc_co_u32 %7, vcc, 0, %7, vcc"
        : "+v"(r1[0]), "+v"(r1[1]), "+v"(r1[2]), "+v"(r1[3]), "+v"(r1[4]), "+v"(r1[5]), "+v"(r1[6]), "+v"(co)
        : "v"(a[8]), "v"(a9)
        : "vcc");
    k += co;
  }
#pragma unroll
  for (int i = 0; i < 7; i++) r.v[i + 1] = r1[i];
  fq_fold_small(r, k);
  return r;
#else
  uint32_t a[10];
  // a = lo + hi*977
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)t[8 + i] * 977u + t[i];
    a[i] = (uint32_t)c;
    c >>= 32;
  }
  a[8] = (uint32_t)c;
  // a += hi << 32
  c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (uint64_t)a[i + 1] + t[8 + i];
    a[i + 1] = (uint32_t)c;
    c >>= 32;
  }
  a[9] = (uint32_t)c;
  // second fold: h2 = a[9]:a[8] (< 2^34), add h2*C = h2*977 + (h2<<32)
  u256 r;
  c = (uint64_t)a[8] * 977u + a[0];
  r.v[0] = (uint32_t)c;
  c >>= 32;
  c += (uint64_t)a[9] * 977u + a[1];
  c += a[8];
  r.v[1] = (uint32_t)c;
  c >>= 32;
  c += (uint64_t)a[2] + a[9];
  r.v[2] = (uint32_t)c;
  c >>= 32;
#pragma unroll
  for (int i = 3; i < 8; i++) {
    c += a[i];
    r.v[i] = (uint32_t)c;
    c >>= 32;
  }
  fq_fold_small(r, (uint32_t)c);
  return r;
#endif
}

BP_HD u256 fq_mul(const u256& a, const u256& b) {
  uint32_t t[16];
  mul_wide(t, a, b);
  return fq_reduce_wide(t);
}
BP_HD u256 fq_sqr(const u256& a) {
  uint32_t t[16];
  sqr_wide(t, a);
  return fq_reduce_wide(t);
}

// unique representative in [0, p)
BP_HD u256 fq_canon(const u256& a) {
  u256 t;
  uint32_t bw = u256_sub(t, a, fq_p());
  return bw ? a : t;
}
// a == 0 (mod p) for weakly reduced a: a in {0, p}
BP_HD bool fq_is_zero(const u256& a) {
  uint32_t z = 0, q = 0;
  z |= a.v[0];
  q |= a.v[0] ^ 0xFFFFFC2Fu;
  z |= a.v[1];
  q |= a.v[1] ^ 0xFFFFFFFEu;
#pragma unroll
  for (int i = 2; i < 8; i++) {
    z |= a.v[i];
    q |= ~a.v[i];
  }
  return z == 0 || q == 0;
}
BP_HD bool fq_eq(const u256& a, const u256& b) { return fq_is_zero(fq_sub(a, b)); }

// a^(2^n)
BP_HD u256 fq_sqr_n(u256 a, int n) {
  for (int i = 0; i < n; i++) a = fq_sqr(a);
  return a;
}

#if defined(__HIP_DEVICE_COMPILE__)
// Per-lane masks in an SGPR pair and selects through them: the compiler turns 8-limb ternaries into
// exec-masked branches full of moves, these stay one v_cndmask per limb.
typedef unsigned long long lanemask;
__device__ __forceinline__ uint32_t vsel(lanemask m, uint32_t a, uint32_t b) {  // m ? a : b
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, %2, %1, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
  return r;
}
__device__ __forceinline__ lanemask vmask_even(uint32_t x) {
  lanemask m;
  uint32_t t = x & 1u;
  asm("v_cmp_eq_u32_e64 %0, 0, %1" : "=s"(m) : "v"(t));
  return m;
}
// d = a - b; returns the lanes that borrowed (a < b)
__device__ __forceinline__ lanemask dev_sub8_mask(uint32_t (&d)[8], const uint32_t (&a)[8], const uint32_t (&b)[8]) {
  lanemask m;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
      "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
      "s_mov_b64 %8, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=s"(m)
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]), "v"(b[1]),
        "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
      : "vcc");
  return m;
}
// d = a + b (no carry out kept)
__device__ __forceinline__ void dev_add8_to(uint32_t (&d)[8], const uint32_t (&a)[8], const uint32_t (&b)[8]) {
  asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
      "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]), "v"(b[1]),
        "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
      : "vcc");
}
#endif

// Kaliski's almost Montgomery inverse for an odd prime m with 2^255 < m < 2^256 and a in [1, m):
// returns a^-1 * 2^k mod m and k, 256 <= k <= 512.  Only shifts, adds and selects on 8 limbs
// (about 120 instructions per step, k = 362 steps on average), against 256 squarings + up to
// 256 multiplications of a Fermat inversion.  Every step keeps u*s + v*r = m with u, v >= 1, so r and s
// stay <= m and fit 8 limbs; the classical loop's last step (u = v = 1) is done after the loop.
BP_HD u256 almost_inverse(const u256& a, const u256& m, uint32_t& k_out) {
  u256 u = m, v = a, r = u256_zero(), s = u256_one();
  uint32_t k = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  // the same step as the portable loop below, spelled as selects: ~130 VALU instructions
  for (; k < 512; k++) {
    uint32_t d[8], e[8], x[8], sum[8], keep[8], dbl[8];
    lanemask lt = dev_sub8_mask(d, u.v, v.v);
    if ((d[0] | d[1] | d[2] | d[3] | d[4] | d[5] | d[6] | d[7]) == 0u) break;
    (void)dev_sub8_mask(e, v.v, u.v);
    lanemask ue = vmask_even(u.v[0]), ve = vmask_even(v.v[0]);
    // od = ~(ue | ve): both odd, subtract step; us = ue | (od & ~lt): u (and s) change, else v (and r).
    // Scalar instructions on the masks: C operators on them are compiled as per-lane 64-bit VALU work.
    lanemask od, us;
    asm("s_nor_b64 %0, %3, %4\n\t"
        "s_andn2_b64 %1, %0, %2\n\t"
        "s_or_b64 %1, %1, %3"
        : "=&s"(od), "=&s"(us)
        : "s"(lt), "s"(ue), "s"(ve)
        : "scc");
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = vsel(od, vsel(lt, e[i], d[i]), vsel(ue, u.v[i], v.v[i]));
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t xh = (x[i] >> 1) | (i < 7 ? x[i + 1] << 31 : 0u);
      u.v[i] = vsel(us, xh, u.v[i]);
      v.v[i] = vsel(us, v.v[i], xh);
    }
    dev_add8_to(sum, r.v, s.v);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      keep[i] = vsel(od, sum[i], vsel(us, r.v[i], s.v[i]));
      dbl[i] = vsel(us, s.v[i], r.v[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint32_t d2 = (dbl[i] << 1) | (i > 0 ? dbl[i - 1] >> 31 : 0u);
      r.v[i] = vsel(us, keep[i], d2);
      s.v[i] = vsel(us, d2, keep[i]);
    }
  }
#else
  for (; k < 512; k++) {  // 511 steps suffice for gcd(a, m) = 1; the bound makes every lane leave the loop
    u256 d, e;
    uint32_t lt = u256_sub(d, u, v);
    if (u256_is_zero(d)) break;
    (void)u256_sub(e, v, u);
    bool ue = (u.v[0] & 1u) == 0, ve = (v.v[0] & 1u) == 0;
    bool odd2 = !ue && !ve;                // subtract step
    bool uside = ue || (odd2 && !lt);      // u (and s) change, else v (and r)
    u256 x, sum;
    (void)u256_add(sum, r, s);
#pragma unroll
    for (int i = 0; i < 8; i++) x.v[i] = odd2 ? (lt ? e.v[i] : d.v[i]) : (ue ? u.v[i] : v.v[i]);
    u256 xh, r2, s2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      xh.v[i] = (x.v[i] >> 1) | (i < 7 ? x.v[i + 1] << 31 : 0u);
      r2.v[i] = (r.v[i] << 1) | (i > 0 ? r.v[i - 1] >> 31 : 0u);
      s2.v[i] = (s.v[i] << 1) | (i > 0 ? s.v[i - 1] >> 31 : 0u);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      u.v[i] = uside ? xh.v[i] : u.v[i];
      v.v[i] = uside ? v.v[i] : xh.v[i];
      uint32_t keep_r = odd2 ? sum.v[i] : r.v[i], keep_s = odd2 ? sum.v[i] : s.v[i];
      r.v[i] = uside ? keep_r : r2.v[i];
      s.v[i] = uside ? s2.v[i] : keep_s;
    }
  }
#endif
  // last step: r <- 2r (mod m), k <- k + 1; the inverse is -r
  u256 t, t2;
  uint32_t top = r.v[7] >> 31;
#pragma unroll
  for (int i = 7; i >= 0; i--) t.v[i] = (r.v[i] << 1) | (i > 0 ? r.v[i - 1] >> 31 : 0u);
  uint32_t bw = u256_sub(t2, t, m);
  if (top || !bw) t = t2;
  (void)u256_sub(t2, m, t);
  k_out = k + 1;
  return t2;
}
// 2^e for 0 <= e <= 255
BP_HD u256 u256_pow2(uint32_t e) {
  u256 t;
#pragma unroll
  for (int i = 0; i < 8; i++) t.v[i] = ((uint32_t)i == (e >> 5)) ? (1u << (e & 31u)) : 0u;
  return t;
}

// a^(p-2) by the standard secp256k1 addition chain (255 squarings, 15 mults).
// fq_inv_fermat(0) = 0.
BP_HD u256 fq_inv_fermat(const u256& a) {
  u256 x2 = fq_mul(fq_sqr(a), a);
  u256 x3 = fq_mul(fq_sqr(x2), a);
  u256 x6 = fq_mul(fq_sqr_n(x3, 3), x3);
  u256 x9 = fq_mul(fq_sqr_n(x6, 3), x3);
  u256 x11 = fq_mul(fq_sqr_n(x9, 2), x2);
  u256 x22 = fq_mul(fq_sqr_n(x11, 11), x11);
  u256 x44 = fq_mul(fq_sqr_n(x22, 22), x22);
  u256 x88 = fq_mul(fq_sqr_n(x44, 44), x44);
  u256 x176 = fq_mul(fq_sqr_n(x88, 88), x88);
  u256 x220 = fq_mul(fq_sqr_n(x176, 44), x44);
  u256 x223 = fq_mul(fq_sqr_n(x220, 3), x3);
  // p-2 = 2^256 - 2^32 - 979: x223 | 0 | 22 ones | 0000 | 1 | 0 | 11 | 0 | 1
  u256 t = fq_mul(fq_sqr_n(x223, 23), x22);
  t = fq_mul(fq_sqr_n(t, 5), a);
  t = fq_mul(fq_sqr_n(t, 3), x2);
  t = fq_mul(fq_sqr_n(t, 2), a);
  return t;
}

// 1/a mod p, fq_inv(0) = 0: almost inverse a^-1 2^k, then times 2^(512-k) and 2^-512
BP_HD u256 fq_inv(const u256& a) {
#ifdef BP_INV_FERMAT
  return fq_inv_fermat(a);
#else
  u256 ac = fq_canon(a);
  if (u256_is_zero(ac)) return u256_zero();
  uint32_t k;
  u256 x = almost_inverse(ac, fq_p(), k);
  uint32_t e = 512u - k;  // 0..256
  u256 t = u256_pow2(e & 255u);
  if (e == 256u) {  // 2^256 = 2^32 + 977 (mod p)
    t = u256_zero();
    t.v[0] = 977u;
    t.v[1] = 1u;
  }
  constexpr uint32_t c[8] = {0x0A7244F9u, 0xAC91B0BEu, 0x2A7F3BB9u, 0xB6421533u,
                             0x0C0528C7u, 0x9606E13Du, 0x9F714620u, 0x35C23D44u};  // 2^-512 mod p
  u256 c512;
#pragma unroll
  for (int i = 0; i < 8; i++) c512.v[i] = c[i];
  return fq_mul(fq_mul(x, t), c512);
#endif
}

// a^((p+1)/4): square root candidate (p = 3 mod 4).  Caller checks r^2 == a.
BP_HD u256 fq_sqrt_candidate(const u256& a) {
  u256 x2 = fq_mul(fq_sqr(a), a);
  u256 x3 = fq_mul(fq_sqr(x2), a);
  u256 x6 = fq_mul(fq_sqr_n(x3, 3), x3);
  u256 x9 = fq_mul(fq_sqr_n(x6, 3), x3);
  u256 x11 = fq_mul(fq_sqr_n(x9, 2), x2);
  u256 x22 = fq_mul(fq_sqr_n(x11, 11), x11);
  u256 x44 = fq_mul(fq_sqr_n(x22, 22), x22);
  u256 x88 = fq_mul(fq_sqr_n(x44, 44), x44);
  u256 x176 = fq_mul(fq_sqr_n(x88, 88), x88);
  u256 x220 = fq_mul(fq_sqr_n(x176, 44), x44);
  u256 x223 = fq_mul(fq_sqr_n(x220, 3), x3);
  // (p+1)/4 = 2^254 - 2^30 - 244: x223 | 0 | 22 ones | 0000 | 11 | 00
  u256 t = fq_mul(fq_sqr_n(x223, 23), x22);
  t = fq_mul(fq_sqr_n(t, 6), x2);
  t = fq_sqr_n(t, 2);
  return t;
}

// ----------------------------------------------------------------------------
// Generic Montgomery (CIOS), canonical values
// ----------------------------------------------------------------------------
struct FrMod {
  static BP_HD uint32_t m(int i) {
    constexpr uint32_t t[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u,
                               0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    return t[i];
  }
  static BP_HD uint32_t r1(int i) {  // 2^256 mod m
    constexpr uint32_t t[8] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 0x00000001u, 0, 0, 0};
    return t[i];
  }
  static BP_HD uint32_t r2(int i) {  // 2^512 mod m
    constexpr uint32_t t[8] = {0x67D7D140u, 0x896CF214u, 0x0E7CF878u, 0x741496C2u,
                               0x5BCD07C6u, 0xE697F5E4u, 0x81C69BC5u, 0x9D671CD5u};
    return t[i];
  }
  static constexpr uint32_t n0inv = 0x5588B13Fu;  // -m^-1 mod 2^32
};
struct FqMod {
  static BP_HD uint32_t m(int i) {
    constexpr uint32_t t[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                               0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    return t[i];
  }
  static BP_HD uint32_t r1(int i) {
    constexpr uint32_t t[8] = {0x000003D1u, 0x00000001u, 0, 0, 0, 0, 0, 0};
    return t[i];
  }
  static BP_HD uint32_t r2(int i) {
    constexpr uint32_t t[8] = {0x000E90A1u, 0x000007A2u, 0x00000001u, 0, 0, 0, 0, 0};
    return t[i];
  }
  static constexpr uint32_t n0inv = 0xD2253531u;
};

template <class M>
struct Mont {
  static BP_HD u256 modulus() {
    u256 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = M::m(i);
    return r;
  }
  static BP_HD u256 one() {
    u256 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = M::r1(i);
    return r;
  }
  static BP_HD u256 add(const u256& a, const u256& b) {
    u256 s, d;
    uint32_t cy = u256_add(s, a, b);
    uint32_t bw = u256_sub(d, s, modulus());
    return (cy || !bw) ? d : s;
  }
  static BP_HD u256 sub(const u256& a, const u256& b) {
    u256 d, e;
    uint32_t bw = u256_sub(d, a, b);
    u256_add(e, d, modulus());
    return bw ? e : d;
  }
  static BP_HD u256 neg(const u256& a) { return sub(u256_zero(), a); }
  // Montgomery product a*b*2^-256 mod m
  static BP_HD u256 mul(const u256& a, const u256& b) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      uint64_t c = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        c += (uint64_t)a.v[i] * b.v[j] + t[j];
        t[j] = (uint32_t)c;
        c >>= 32;
      }
      c += t[8];
      t[8] = (uint32_t)c;
      t[9] = (uint32_t)(c >> 32);
      uint32_t q = t[0] * M::n0inv;
      c = (uint64_t)q * M::m(0) + t[0];
      c >>= 32;
#pragma unroll
      for (int j = 1; j < 8; j++) {
        c += (uint64_t)q * M::m(j) + t[j];
        t[j - 1] = (uint32_t)c;
        c >>= 32;
      }
      c += t[8];
      t[7] = (uint32_t)c;
      t[8] = t[9] + (uint32_t)(c >> 32);
    }
    u256 r, d;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    uint32_t bw = u256_sub(d, r, modulus());
    return (t[8] || !bw) ? d : r;
  }
  static BP_HD u256 sqr(const u256& a) { return mul(a, a); }
  static BP_HD u256 to_mont(const u256& a) {
    u256 r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = M::r2(i);
    return mul(a, r2);
  }
  static BP_HD u256 from_mont(const u256& a) { return mul(a, u256_one()); }
  // a^e, e given as plain 256-bit integer (square-and-multiply, MSB first)
  static BP_HD u256 pow(const u256& a, const u256& e) {
    u256 r = one(), ee = e;
    for (int i = 255; i >= 0; i--) {
      r = sqr(r);
      if (u256_pop_msb(ee)) r = mul(r, a);
    }
    return r;
  }
  // a^(m-2); inv(0) = 0
  static BP_HD u256 inv_fermat(const u256& a) {
    u256 e = modulus();
    e.v[0] -= 2;  // low limb of both moduli is >= 2
    return pow(a, e);
  }
  // Montgomery inverse: a = A R  ->  A^-1 R, inv(0) = 0.  almost_inverse gives a^-1 2^k; one Montgomery
  // product with 2^(512-k) makes it a^-1 R, one with R^2 makes it a^-1 R^2 = A^-1 R.
  static BP_HD u256 inv(const u256& a) {
#ifdef BP_INV_FERMAT
    return inv_fermat(a);
#else
    if (u256_is_zero(a)) return u256_zero();
    uint32_t k;
    u256 x = almost_inverse(a, modulus(), k);
    uint32_t e = 512u - k;  // 0..256
    u256 t = u256_pow2(e & 255u);
    if (e == 256u) t = one();
    u256 r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = M::r2(i);
    return mul(mul(x, t), r2);
#endif
  }
  // reduce an arbitrary 256-bit integer (< 2^256 < 2m) to canonical
  static BP_HD u256 reduce_once(const u256& a) {
    u256 d;
    uint32_t bw = u256_sub(d, a, modulus());
    return bw ? a : d;
  }
};

typedef Mont<FrMod> Fr;
typedef Mont<FqMod> FqMont;

}  // namespace bp
