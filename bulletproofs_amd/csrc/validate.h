// Host-side checks of caller-supplied encodings, shared by the verifiers that take untrusted proofs
// (bp_rangeproof_verify, bp_ipp_verify, bp_ac_verify).  The reference's types cannot hold such values: an `Fr` / `Fq`
// is reduced by construction and a `PA` built with `A x y` off the curve would simply run through the affine formulas
// (/root/reference/Bulletproofs/Utils.hs:8-9 imports them from elliptic-curve / galois-field).  These verifiers answer
// "does not verify" for them, like the batch verifier's k_vcheck (verify.hip): the only answer that cannot be turned
// into a false accept.
#pragma once
#include <stdint.h>
#include <string.h>

#include "ec.h"
#include "host_fr.h"

namespace bp {
namespace host {

static const uint64_t VAL_P[4] = {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                                  0xFFFFFFFFFFFFFFFFULL};

static inline bool le_below(const uint8_t le[32], const uint64_t m[4]) {
  uint64_t v[4];
  memcpy(v, le, 32);
  for (int i = 3; i >= 0; i--) {
    if (v[i] < m[i]) return true;
    if (v[i] > m[i]) return false;
  }
  return false;
}

// O (64 zero bytes), or canonical coordinates of a point on y^2 = x^3 + 7
static inline bool point_encoding_valid(const uint8_t pt[64]) {
  affine a;
  memcpy(&a, pt, 64);
  if (affine_is_inf(a)) return true;
  if (!le_below(pt, VAL_P) || !le_below(pt + 32, VAL_P)) return false;
  return affine_on_curve(a);
}
static inline bool points_encoding_valid(const uint8_t* pts, size_t n) {
  for (size_t i = 0; i < n; i++)
    if (!point_encoding_valid(pts + i * 64)) return false;
  return true;
}
// a canonical scalar (< r)
static inline bool scalar_encoding_valid(const uint8_t s[32]) { return le_below(s, FR_M); }

}  // namespace host
}  // namespace bp
