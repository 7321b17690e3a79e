// Host-side modular inversion for the latency paths: the verifiers hand the field inversions of a single proof
// (x_r^-1 mod r, the normalisation 1/ZZZ mod p of each running commitment) to the CPU, where a binary extended
// Euclid on four 64-bit limbs takes a few microseconds, against ~0.14 ms for one GPU lane and ~28 us for the
// portable host build of the device routines (fe.h almost_inverse / Fermat).  Host only.
#pragma once
#include <stdint.h>
#include <string.h>
#include "ec.h"
#include "host_fr.h"

namespace bp {
namespace host {

static const uint64_t INV_P[4] = {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL};

// 1/a mod p for a base-field element in the weak representation of fe.h; 0 for 0
static inline u256 fq_inv_host(const u256& a) {
  u256 c = fq_canon(a), r;
  uint64_t in[4], out[4];
  memcpy(in, c.v, 32);
  modinv256(in, INV_P, out);
  memcpy(r.v, out, 32);
  return r;
}
// x^-1 for a scalar in Montgomery form (R = 2^256, the form of fe.h's Fr and of host_fr.h's Fr4), Montgomery form out
static inline u256 fr_inv_mont_host(const u256& xm) {
  // plain inverse of x R is x^-1 R^-1; two Montgomery products with R^2 bring it to x^-1 R
  u256 c = Fr::reduce_once(xm), r;
  uint64_t in[4], out[4];
  memcpy(in, c.v, 32);
  modinv256(in, FR_M, out);
  memcpy(r.v, out, 32);
  return Fr::to_mont(Fr::to_mont(r));
}
// (X, Y, ZZ, ZZZ) -> affine with canonical coordinates; the identity gives (0, 0)
static inline affine xyzz_to_affine_host(const xyzz& p) {
  if (xyzz_is_inf(p)) return affine_inf();
  return xyzz_to_affine_with_inv(p, fq_inv_host(p.ZZZ));
}

// n points with ONE inversion (Montgomery's trick over the finite ones)
static inline void xyzz_batch_to_affine_host(const xyzz* p, int n, affine* out) {
  enum { CHUNK = 32 };
  for (int base = 0; base < n; base += CHUNK) {
    const int cnt = n - base < CHUNK ? n - base : CHUNK;
    u256 pre[CHUNK], acc = u256_one();
    for (int i = 0; i < cnt; i++) {
      pre[i] = acc;
      if (!xyzz_is_inf(p[base + i])) acc = fq_mul(acc, p[base + i].ZZZ);
    }
    u256 inv = fq_inv_host(acc);
    for (int i = cnt - 1; i >= 0; i--) {
      const xyzz& q = p[base + i];
      if (xyzz_is_inf(q)) {
        out[base + i] = affine_inf();
        continue;
      }
      out[base + i] = xyzz_to_affine_with_inv(q, fq_mul(inv, pre[i]));
      inv = fq_mul(inv, q.ZZZ);
    }
  }
}

}  // namespace host
}  // namespace bp
